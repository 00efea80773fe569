// Does the VGPR NUMBERING of an FP64 instruction's operands change what a lone wave pays for it on gfx950? (The step loops
// of the wave-per-agent kernels came out +-1 % between builds whose loops differ in register numbers only.)
// One wave per block, 64 blocks; 8 independent v_fma_f64 per iteration with hand-picked registers:
//   A  sources in three different register pairs modulo 4 where possible: v[0:1] v[2:3] + accumulators
//   B  all three sources congruent modulo 4 (v[0:1], v[4:5], v[8:9] ...)
//   C  sources congruent modulo 8 / 16 / 32
// Prints shader-clock ticks per instruction. Build: hipcc --offload-arch=gfx950 -O3 tools/vgprbank.hip -o tools/vgprbank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 8192
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
#define FMA(d, a, b, c) "v_fma_f64 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #c ":" #c "+1]\n\t"
template <int V>
__global__ __launch_bounds__(64) void k(unsigned long long *cyc, double *out) {
  // registers v0..v95 initialised to small values so the FMAs stay finite (x = x * 1 + 0 pattern)
  asm volatile(
      "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0x3ff00000\n\t"      // v[0:1] = 1.0
      "v_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t"               // v[2:3] = 0.0
      ::: "v0", "v1", "v2", "v3");
#define INIT1(r) asm volatile("v_mov_b32 v" #r ", 0" ::: "v" #r);
  INIT1(4) INIT1(5) INIT1(6) INIT1(7) INIT1(8) INIT1(9) INIT1(10) INIT1(11) INIT1(12) INIT1(13) INIT1(14) INIT1(15)
  INIT1(16) INIT1(17) INIT1(18) INIT1(19) INIT1(20) INIT1(21) INIT1(22) INIT1(23) INIT1(24) INIT1(25) INIT1(26) INIT1(27)
  INIT1(28) INIT1(29) INIT1(30) INIT1(31) INIT1(32) INIT1(33) INIT1(34) INIT1(35) INIT1(36) INIT1(37) INIT1(38) INIT1(39)
  INIT1(40) INIT1(41) INIT1(42) INIT1(43) INIT1(44) INIT1(45) INIT1(46) INIT1(47) INIT1(48) INIT1(49) INIT1(50) INIT1(51)
  INIT1(52) INIT1(53) INIT1(54) INIT1(55) INIT1(56) INIT1(57) INIT1(58) INIT1(59) INIT1(60) INIT1(61) INIT1(62) INIT1(63)
  INIT1(64) INIT1(65) INIT1(66) INIT1(67) INIT1(68) INIT1(69) INIT1(70) INIT1(71) INIT1(72) INIT1(73) INIT1(74) INIT1(75)
  INIT1(76) INIT1(77) INIT1(78) INIT1(79) INIT1(80) INIT1(81) INIT1(82) INIT1(83) INIT1(84) INIT1(85) INIT1(86) INIT1(87)
  INIT1(88) INIT1(89) INIT1(90) INIT1(91) INIT1(92) INIT1(93) INIT1(94) INIT1(95)
  unsigned long long t0 = now();
  for (int i = 0; i < N_IT; i++) {
    if (V == 0)   // sources v[0:1] (x1), v[2:3] (x2): two different pairs mod 4; destinations spread
      asm volatile(FMA(16, 16, 0, 2) FMA(22, 22, 0, 2) FMA(28, 28, 0, 2) FMA(34, 34, 0, 2)
                   FMA(40, 40, 0, 2) FMA(46, 46, 0, 2) FMA(52, 52, 0, 2) FMA(58, 58, 0, 2) :::);
    if (V == 1)   // all three sources = 0 mod 4
      asm volatile(FMA(16, 16, 0, 4) FMA(20, 20, 0, 4) FMA(24, 24, 0, 4) FMA(28, 28, 0, 4)
                   FMA(32, 32, 0, 4) FMA(36, 36, 0, 4) FMA(40, 40, 0, 4) FMA(44, 44, 0, 4) :::);
    if (V == 2)   // sources 0, 2 mod 4 and accumulator 2 mod 4
      asm volatile(FMA(18, 18, 0, 2) FMA(22, 22, 0, 2) FMA(26, 26, 0, 2) FMA(30, 30, 0, 2)
                   FMA(34, 34, 0, 2) FMA(38, 38, 0, 2) FMA(42, 42, 0, 2) FMA(46, 46, 0, 2) :::);
    if (V == 3)   // all three sources = 0 mod 8
      asm volatile(FMA(16, 16, 0, 8) FMA(24, 24, 0, 8) FMA(32, 32, 0, 8) FMA(40, 40, 0, 8)
                   FMA(48, 48, 0, 8) FMA(56, 56, 0, 8) FMA(64, 64, 0, 8) FMA(72, 72, 0, 8) :::);
    if (V == 4)   // odd-aligned pairs: v[1:2]-style operands are not allowed for 64-bit; use sources 0 mod 4 and 2 mod 4, accumulators 0 mod 4
      asm volatile(FMA(16, 16, 2, 6) FMA(20, 20, 2, 6) FMA(24, 24, 2, 6) FMA(28, 28, 2, 6)
                   FMA(32, 32, 2, 6) FMA(36, 36, 2, 6) FMA(40, 40, 2, 6) FMA(44, 44, 2, 6) :::);
    if (V == 5)   // same source register twice (x * x + c)
      asm volatile(FMA(16, 0, 0, 16) FMA(20, 0, 0, 20) FMA(24, 0, 0, 24) FMA(28, 0, 0, 28)
                   FMA(32, 0, 0, 32) FMA(36, 0, 0, 36) FMA(40, 0, 0, 40) FMA(44, 0, 0, 44) :::);
  }
  unsigned long long t1 = now();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  double r;
  asm volatile("v_mov_b32 %0, v16" : "=v"(*reinterpret_cast<int *>(&r)));
  out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int V> void run(const char *name) {
  double *out; unsigned long long *cyc;
  hipMalloc(&out, 64 * 64 * 8); hipMalloc(&cyc, 64 * 8);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<V>), dim3(64), dim3(64), 0, 0, cyc, out);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(64); hipMemcpy(h.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v; s /= 64;
  printf("%-64s %7.3f memtime ticks per v_fma_f64\n", name, s / N_IT / 8);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("acc 0 mod 2 spread, sources v[0:1] v[2:3]");
  run<1>("acc, both sources all 0 mod 4");
  run<2>("acc 2 mod 4, sources 0 and 2 mod 4");
  run<3>("acc, both sources all 0 mod 8");
  run<4>("acc 0 mod 4, sources 2 mod 4 (v[2:3], v[6:7])");
  run<5>("x * x + acc, x = v[0:1], acc 0 mod 4");
  return 0;
}

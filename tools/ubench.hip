// Micro-benchmarks of the FP64 building blocks of the rollout kernel, one wave
// per block, 64 blocks (the C2 shape). Prints cycles per operation from
// s_memtime (shader clock). Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 4096
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
template <int OP, int ACTIVE = 64>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, double seed) {
  if ((int)threadIdx.x >= ACTIVE) return;  // EXEC keeps only the low ACTIVE lanes for the whole kernel
  double a = seed + threadIdx.x * 1e-3, b = 1.0000001 + threadIdx.x * 1e-9, c = 0.5, d = seed * 0.3, e = seed * 0.7;
  unsigned long long t0 = now();
  for (int i = 0; i < N_IT; i++) {
    if (OP == 0) { a = a * b + c; }                                  // dependent mul+add (2 instr, no contraction)
    if (OP == 1) { a = a * b; c = c * b; d = d * b; e = e * b; }     // 4 independent muls
    if (OP == 2) { a = a / b; }                                      // dependent division
    if (OP == 3) { a = a / b; c = c / b; d = d / b; }                // 3 independent divisions
    if (OP == 4) { a = __builtin_sqrt(a) + c; }                      // dependent sqrt + add
    if (OP == 5) { a = __builtin_sqrt(a) + c; d = __builtin_sqrt(d) + c; e = __builtin_sqrt(e) + c; }
    if (OP == 6) { int lo = __double2loint(a), hi = __double2hiint(a); lo = __builtin_amdgcn_readlane(lo, i & 63); hi = __builtin_amdgcn_readlane(hi, i & 63); a = __hiloint2double(hi, lo) + c; }
    if (OP == 7) { double o = __shfl_xor(a, 16); a = (o < a ? o : a) + c; }
    if (OP == 8) { int lo = __double2loint(a), hi = __double2hiint(a); lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); double o = __hiloint2double(hi, lo); a = (o < a ? o : a) + c; }
    if (OP == 9) { if (__builtin_amdgcn_readfirstlane(i) & 1) a = a + c; else a = a * b; }  // uniform branch
    if (OP == 10) { a = __builtin_fma(a, b, c); }                    // dependent fma
    if (OP == 11) { a = a + c; }                                     // dependent add
  }
  unsigned long long t1 = now();
  out[blockIdx.x * 64 + threadIdx.x] = a + c + d + e;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, int ACTIVE = 64> void run(const char *name, int ops) {
  double *out; unsigned long long *cyc;
  hipMalloc(&out, 64 * 64 * 8); hipMalloc(&cyc, 64 * 8);
  hipLaunchKernelGGL((k<OP, ACTIVE>), dim3(64), dim3(64), 0, 0, out, cyc, 1.25);
  hipLaunchKernelGGL((k<OP, ACTIVE>), dim3(64), dim3(64), 0, 0, out, cyc, 1.25);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(64); hipMemcpy(h.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v; s /= 64;
  printf("%-34s %8.1f memtime-ticks/iter  (%.1f per op)\n", name, s / N_IT, s / N_IT / ops);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<11>("dep add", 1); run<10>("dep fma", 1); run<0>("dep mul+add", 2); run<1>("4 indep mul", 4);
  run<2>("dep div", 1); run<3>("3 indep div", 3); run<4>("dep sqrt+add", 1); run<5>("3 indep sqrt+add", 3);
  run<6>("readlane x2 + add", 1); run<7>("shfl_xor(bpermute) min + add", 1); run<8>("dpp x2 min + add", 1);
  run<9>("uniform branch + op", 1);
  // does a partially filled wave issue faster? (EXEC with only the low 32 / 16 lanes set)
  run<1, 32>("4 indep mul, 32 active lanes", 4); run<1, 16>("4 indep mul, 16 active lanes", 4);
  run<10, 32>("dep fma, 32 active lanes", 1); run<10, 16>("dep fma, 16 active lanes", 1);
  run<3, 32>("3 indep div, 32 active lanes", 3); run<5, 32>("3 indep sqrt+add, 32 active lanes", 3);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("clockRate kHz %d\n", clk);
  int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0); printf("wallClockRate kHz %d\n", wc);
  return 0;
}

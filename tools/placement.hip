// where does the dispatcher put the waves of a one-wave-per-block launch? (HW_ID / XCC_ID per wave)
// build: hipcc --offload-arch=gfx950 -O2 tools/placement.hip -o tools/placement ; run: tools/placement <blocks> <vgpr:0|1> <lds bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
template <int BIG>
__global__ __launch_bounds__(64) void k(unsigned *out, unsigned long long *t, int spin) {
  extern __shared__ double smem[];
  if (BIG) asm volatile("v_mov_b32 v250, 0" ::: "v250");
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  const unsigned long long t0 = wall_clock64();
  smem[threadIdx.x] = (double)hw;
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc & 15u; t[blockIdx.x] = t0; }
}
int main(int argc, char **argv) {
  int blocks = atoi(argv[1]), big = atoi(argv[2]), lds = atoi(argv[3]);
  unsigned *d; unsigned long long *dt;
  hipMalloc(&d, blocks * 8); hipMalloc(&dt, blocks * 8);
  for (int rep = 0; rep < 2; rep++) {
    if (big) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), lds, 0, d, dt, 20000);
    else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), lds, 0, d, dt, 20000);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(2 * blocks); std::vector<unsigned long long> ht(blocks);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, blocks * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_simd, per_cu; std::map<unsigned, std::vector<int>> who;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int b = 0; b < blocks; b++) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1];
    unsigned simd = (hw >> 4) & 3, cu_se = (hw >> 8) & 0xff;   // cu_id[11:8] sh_id[12] se_id[15:13]
    unsigned key_cu = (xcc << 8) | cu_se, key = (key_cu << 2) | simd;
    per_simd[key]++; per_cu[key_cu]++; who[key].push_back(b);
    if (ht[b] < tmin) tmin = ht[b]; if (ht[b] > tmax) tmax = ht[b];
  }
  std::map<int, int> hist, histcu;
  for (auto &p : per_simd) hist[p.second]++;
  for (auto &p : per_cu) histcu[p.second]++;
  printf("blocks %d big %d lds %d: SIMDs used %zu, CUs used %zu, start spread %llu ticks\n", blocks, big, lds, per_simd.size(), per_cu.size(), tmax - tmin);
  printf("  waves per SIMD histogram:"); for (auto &p : hist) printf(" %d:%d", p.first, p.second); printf("\n");
  printf("  waves per CU histogram:"); for (auto &p : histcu) printf(" %d:%d", p.first, p.second); printf("\n");
  int shown = 0;
  for (auto &p : who) if (p.second.size() >= 2 && shown++ < 6) { printf("  simd %x blocks:", p.first); for (int b : p.second) printf(" %d", b); printf("\n"); }
  printf("  first 16 blocks (xcc.cu_se.simd):"); for (int b = 0; b < 16 && b < blocks; b++) printf(" %u.%02x.%u", h[2*b+1], (h[2*b] >> 8) & 0xff, (h[2*b] >> 4) & 3); printf("\n");
  return 0;
}

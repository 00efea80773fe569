// where do the W waves of a block go? (HW_ID per wave): distinct SIMDs per block
// build: hipcc --offload-arch=gfx950 -O2 tools/mwplace.hip -o tools/mwplace ; run: tools/mwplace <blocks> <waves> <lds bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
__global__ void k(unsigned *out, int spin) {
  extern __shared__ double smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  const unsigned long long t0 = wall_clock64();
  smem[threadIdx.x] = (double)hw;
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  const int W = blockDim.x / 64, w = threadIdx.x / 64;
  if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * W + w)] = hw; out[2 * (blockIdx.x * W + w) + 1] = xcc & 15u; }
}
int main(int argc, char **argv) {
  int blocks = atoi(argv[1]), W = atoi(argv[2]), lds = atoi(argv[3]);
  unsigned *d; hipMalloc(&d, blocks * W * 8);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * W), lds, 0, d, 20000); hipDeviceSynchronize(); }
  std::vector<unsigned> h(2 * blocks * W); hipMemcpy(h.data(), d, blocks * W * 8, hipMemcpyDeviceToHost);
  std::map<int, int> distinct; std::map<unsigned, int> per_cu;
  for (int b = 0; b < blocks; b++) {
    std::set<unsigned> simds; unsigned cu = 0;
    for (int w = 0; w < W; w++) { unsigned hw = h[2 * (b * W + w)], xcc = h[2 * (b * W + w) + 1]; simds.insert((hw >> 4) & 3); cu = (xcc << 8) | ((hw >> 8) & 0xff); }
    distinct[(int)simds.size()]++; per_cu[cu]++;
  }
  std::map<int, int> hist; for (auto &p : per_cu) hist[p.second]++;
  printf("blocks %d waves %d lds %d: distinct SIMDs per block:", blocks, W, lds); for (auto &p : distinct) printf(" %d:%d", p.first, p.second);
  printf(" | blocks per CU:"); for (auto &p : hist) printf(" %d:%d", p.first, p.second);
  printf(" | block 0 simds:"); for (int w = 0; w < W; w++) printf(" %u", (h[2 * w] >> 4) & 3); printf("\n");
  return 0;
}

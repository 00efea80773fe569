// exec_rows.hip -- does a VALU FP64 instruction cost fewer issue cycles when only one 16-lane row of the wave is
// active? (If it did, the wave-uniform part of the rollout step could run under a one-row exec mask.)
// One wave; N independent fma streams (issue-bound) and one dependent chain (latency-bound), exec = 64 / 32 / 16 / 1 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int active, int iters, double *out, unsigned long long *cyc) {
  const int lane = threadIdx.x;
  double a0 = 1.0 + lane * 1e-9, a1 = a0 + 1e-3, a2 = a0 + 2e-3, a3 = a0 + 3e-3, a4 = a0 + 4e-3, a5 = a0 + 5e-3, a6 = a0 + 6e-3, a7 = a0 + 7e-3;
  const double m = 1.0000001, c = 1e-9;
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (lane < active) {
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {   // 8 independent chains: issue-bound
      a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {   // one dependent chain: latency-bound
      a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
      a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
    }
    t2 = __builtin_readcyclecounter();
  }
  out[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  double *out; unsigned long long *cyc, h[2];
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
  const int iters = 20000;
  for (int rep = 0; rep < 2; rep++)
    for (int active : {64, 48, 32, 16, 8, 1}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, active, iters, out, cyc);
      hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      printf("active lanes %2d: independent fma %.2f cycles/instr, dependent fma %.2f cycles/instr (shader clock counter)\n", active,
             (double)h[0] / (8.0 * iters), (double)h[1] / (8.0 * iters));
    }
  return 0;
}

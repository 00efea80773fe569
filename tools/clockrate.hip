// Core clock of the GPU while a lone wave computes: s_memtime (core-clock counter) against s_memrealtime (the constant
// 100 MHz wall clock) around a busy loop. Used by tools/slackprof.py to turn nanoseconds into 4-cycle issue slots.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/clockrate.hip -o /tmp/clockrate && /tmp/clockrate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *out, double *sink, int n) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; i++) a = a * b + 1e-9;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[threadIdx.x] = a;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
  unsigned long long *d, h[128];
  double *s;
  hipMalloc(&d, sizeof(h)); hipMalloc(&s, 64 * 8);
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d, s, 2000000);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double tc = 0, tr = 0;
  for (int i = 0; i < 64; i++) { tc += (double)h[2 * i]; tr += (double)h[2 * i + 1]; }
  printf("core_clock_ghz %.4f\n", tc / tr * 0.1);
  return 0;
}

// calib_traffic.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on KNOWN byte counts in the access patterns of
// the rollout kernel (MI355X_MICROARCH.md: the x2 correction of FETCH_SIZE is established for wide coalesced 16 B/lane
// streams only; "calibrate on a known byte count in your own access pattern before trusting an absolute").
//   k_path_pattern   what k_rollout_w64 does to its path: per step three 8-byte stores of a wave-uniform value (all
//                    lanes, same address), then -- after an agent-scope fence -- the cost pass's agent-scope 8-byte
//                    loads, 2 x 3 per point (q and q_prev). 64 waves x 201 points = C2. Known: 308 736 B written,
//                    617 472 B loaded per launch.
//   k_stream_read8 / k_stream_read16 / k_stream_write8   coalesced streams over 64 MiB (beyond the 4 MiB L2).
// Build + run (tools/calib_traffic.sh): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ double ld_agent(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64) void k_path_pattern(double *paths, int cap, double *sink) {
  const int lane = threadIdx.x;
  double *path = paths + (size_t)blockIdx.x * cap * 3;
  double x = 0.001 * blockIdx.x;
  for (int n = 0; n < cap; n++) {  // the step loop's stores: every lane, same address, wave-uniform value
    x = x * 1.0000001 + 1e-3;
    path[n * 3] = x; path[n * 3 + 1] = x + 1.0; path[n * 3 + 2] = x + 2.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  double acc = 0.0;
  for (int base = 0; base < cap; base += 64) {  // path_cost_terms_w64's loads
    const int k = base + lane;
    const bool valid = k < cap;
    const int kk = valid ? k : 0, kp = (valid && k > 0) ? (k - 1) : 0;
    acc += ld_agent(path + kk * 3) + ld_agent(path + kk * 3 + 1) + ld_agent(path + kk * 3 + 2);
    acc += ld_agent(path + kp * 3) + ld_agent(path + kp * 3 + 1) + ld_agent(path + kp * 3 + 2);
  }
  if (acc == 12345.678) sink[0] = acc;
}

__global__ void k_stream_read8(const double *src, size_t n, double *sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc == 12345.678) sink[0] = acc;
}
__global__ void k_stream_read16(const double2 *src, size_t n, double *sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = src[i]; acc += v.x + v.y; }
  if (acc == 12345.678) sink[0] = acc;
}
__global__ void k_stream_write8(double *dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)i;
}

int main() {
  const int waves = 64, cap = 201, reps = 20;
  const size_t big = 64u << 20;  // bytes
  double *paths, *buf, *sink;
  CHECK(hipMalloc(&paths, sizeof(double) * waves * cap * 3));
  CHECK(hipMalloc(&buf, big));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 0, big));
  for (int r = 0; r < reps; r++) {
    hipLaunchKernelGGL(k_path_pattern, dim3(waves), dim3(64), 0, 0, paths, cap, sink);
    hipLaunchKernelGGL(k_stream_read8, dim3(2048), dim3(256), 0, 0, buf, big / 8, sink);
    hipLaunchKernelGGL(k_stream_read16, dim3(2048), dim3(256), 0, 0, (const double2 *)buf, big / 16, sink);
    hipLaunchKernelGGL(k_stream_write8, dim3(2048), dim3(256), 0, 0, buf, big / 8);
  }
  CHECK(hipDeviceSynchronize());
  printf("known bytes per launch: k_path_pattern stores %d, agent-scope loads %d; k_stream_* %zu\n",
         waves * cap * 24, 2 * waves * cap * 24, big);
  return 0;
}

// graphlaunch.hip -- what one tick's enqueue costs on this runtime, three ways (round 4):
//   A  two stream launches (short "manager" kernel -> long "rollout" kernel), as pmaf_tick does today
//   B  the same pair captured once as a hipGraph and replayed with hipGraphLaunch; the per-tick arguments (sequence
//      number) are read from mapped pinned memory, so the graph never needs new node parameters
//   C  ONE launch: block 0 does the manager's part and releases a device flag, the other blocks wait for the flag
// Per mode, 400 ticks on an idle stream: host time of the enqueue, host call -> sequence number in pinned memory
// (the set-point latency), and the device-side gap manager end -> rollout start (device wall clock, 100 MHz).
// build: hipcc --offload-arch=gfx950 -O2 tools/graphlaunch.hip -o tools/graphlaunch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct TickArgs { double seq; };
__device__ __forceinline__ void spin(unsigned long long ticks, double *sink) {
  const unsigned long long t0 = wall_clock64();
  double a = threadIdx.x;
  while (wall_clock64() - t0 < ticks) a = a * 1.0000001 + 1e-9;
  if (a == 12345.0) sink[0] = a;
}
__global__ void k_mgr(const volatile TickArgs *A, volatile double *mbox, unsigned long long *stamp, double *sink) {
  spin(900, sink);                                   // ~9 us
  if (threadIdx.x == 0) { mbox[1] = 1.0; __threadfence_system(); mbox[0] = A->seq; stamp[0] = wall_clock64(); }
}
__global__ void k_roll(unsigned long long *stamp, double *sink) {
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
  spin(20000, sink);                                 // ~200 us
}
__global__ void k_fused(const volatile TickArgs *A, volatile double *mbox, unsigned long long *stamp, double *sink, double *flag) {
  if (blockIdx.x == 0) {
    spin(900, sink);
    if (threadIdx.x == 0) {
      mbox[1] = 1.0; __threadfence_system(); mbox[0] = A->seq; stamp[0] = wall_clock64();
      __hip_atomic_store(flag, A->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != A->seq) __builtin_amdgcn_s_sleep(1);
  if (threadIdx.x == 0 && blockIdx.x == 1) stamp[1] = wall_clock64();
  spin(20000, sink);
}
static void stats(const char *w, std::vector<double> &v) {
  std::sort(v.begin(), v.end());
  printf("  %-34s median %7.2f  p90 %7.2f  p99 %7.2f  max %7.2f us\n", w, v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back());
}
int main() {
  hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  TickArgs *hA, *dA; double *hM, *dM, *sink, *flag; unsigned long long *dS, hS[2];
  CHECK(hipHostMalloc((void **)&hA, sizeof(TickArgs), hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&dA, hA, 0));
  CHECK(hipHostMalloc((void **)&hM, 16, hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&dM, hM, 0));
  CHECK(hipMalloc(&sink, 8)); CHECK(hipMalloc(&flag, 8)); CHECK(hipMemset(flag, 0, 8)); CHECK(hipMalloc(&dS, 16));
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_mgr, dim3(1), dim3(64), 0, s, dA, hM ? dM : dM, dS, sink);
  hipLaunchKernelGGL(k_roll, dim3(64), dim3(64), 0, s, dS, sink);
  CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  double seq = 0;
  for (int mode = 0; mode < 3; mode++) {
    std::vector<double> enq, sp, gap;
    for (int it = 0; it < 420; it++) {
      CHECK(hipStreamSynchronize(s));
      seq += 1.0; hA->seq = seq;
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) {
        hipLaunchKernelGGL(k_mgr, dim3(1), dim3(64), 0, s, dA, dM, dS, sink);
        hipLaunchKernelGGL(k_roll, dim3(64), dim3(64), 0, s, dS, sink);
      } else if (mode == 1) {
        CHECK(hipGraphLaunch(ge, s));
      } else {
        hipLaunchKernelGGL(k_fused, dim3(65), dim3(64), 0, s, dA, dM, dS, sink, flag);
      }
      const auto t1 = std::chrono::steady_clock::now();
      while (*(volatile double *)hM != seq) {}
      const auto t2 = std::chrono::steady_clock::now();
      CHECK(hipStreamSynchronize(s));
      CHECK(hipMemcpy(hS, dS, 16, hipMemcpyDeviceToHost));
      if (it >= 20) {
        enq.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        sp.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
        gap.push_back(((double)hS[1] - (double)hS[0]) * 0.01);
      }
    }
    printf("%s\n", mode == 0 ? "A two stream launches" : mode == 1 ? "B hipGraphLaunch (2 kernel nodes)" : "C one fused launch (flag hand-off)");
    stats("host enqueue", enq); stats("host call -> seq in pinned memory", sp); stats("device gap manager end -> rollout start", gap);
  }
  return 0;
}

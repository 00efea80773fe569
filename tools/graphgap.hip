// graphgap.hip -- idle time between two dependent kernels (a short one and a long one, like k_manager -> k_rollout ->
// k_manager ...) when they are launched on a stream vs. replayed as a hipGraph; device wall clock (100 MHz) read at
// the start and the end of every kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_work(unsigned long long *stamps, int slot, int spin_ticks, double *sink) {
  const unsigned long long t0 = wall_clock64();
  double a = threadIdx.x;
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) a = a * 1.0000001 + 1e-9;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[2 * slot] = t0; stamps[2 * slot + 1] = wall_clock64(); }
  if (a == 12345.0) sink[0] = a;
}
static void report(const char *what, std::vector<unsigned long long> &h, int n) {
  std::vector<double> g_ab, g_ba;
  for (int i = 0; i + 1 < n; i++) {
    double gap = (double)(h[2 * (i + 1)] - h[2 * i + 1]) * 0.01;   // 100 MHz -> us
    (i % 2 == 0 ? g_ab : g_ba).push_back(gap);
  }
  std::sort(g_ab.begin(), g_ab.end()); std::sort(g_ba.begin(), g_ba.end());
  printf("%-28s gap short->long median %.2f us, long->short median %.2f us\n", what, g_ab[g_ab.size() / 2], g_ba[g_ba.size() / 2]);
}
int main() {
  const int pairs = 200, n = 2 * pairs;
  unsigned long long *d; double *sink;
  CHECK(hipMalloc(&d, sizeof(unsigned long long) * 2 * n)); CHECK(hipMalloc(&sink, 8));
  std::vector<unsigned long long> h(2 * n);
  hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto launch_all = [&](hipStream_t st) {
    for (int i = 0; i < n; i++) {
      if (i % 2 == 0) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, i, 900, sink);      // ~9 us, 1 wave
      else hipLaunchKernelGGL(k_work, dim3(64), dim3(64), 0, st, d, i, 30000, sink);              // ~300 us, 64 waves
    }
  };
  for (int rep = 0; rep < 2; rep++) {
    launch_all(s); CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost));
    report("stream launches", h, n);
  }
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  launch_all(s);
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 2; rep++) {
    CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost));
    report("hipGraph replay", h, n);
  }
  return 0;
}

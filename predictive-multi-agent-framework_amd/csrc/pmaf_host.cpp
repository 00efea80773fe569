// pmaf_host.cpp -- host side of libpmaf_hip.so: the C-ABI of include/pmaf.h (handle, buffers, streams, the tick
// protocol, getters, checkpointing) over the kernels of pmaf_k_*.hip (launch interface: pmaf_types.hpp).
// Plain C++ against the HIP runtime API; there is no CPU fallback in this library.
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/pmaf.h"
#include "pmaf_comm.hpp"
#include "pmaf_types.hpp"
#include "pmaf_lpa_model.hpp"

#ifndef PMAF_W64_SLICE_DEFAULT
#define PMAF_W64_SLICE_DEFAULT true
#endif

using namespace pmaf;

static thread_local std::string g_err;
void pmaf_set_last_error(const std::string &msg) { g_err = msg; }  // pmaf_shard.cpp reports through the same channel

struct HipError {
  hipError_t e;
  const char *what;
  int line;
};
#define HIP_CHECK(x)                                   \
  do {                                                 \
    hipError_t _e = (x);                               \
    if (_e != hipSuccess) throw HipError{_e, #x, __LINE__}; \
  } while (0)

struct StatusError {
  int code;
  std::string msg;
};
static void fail(int code, const std::string &msg) { throw StatusError{code, msg}; }

// Supported numeric range of every input (metres, m/s, gains, seconds): finite
// and either exactly 0 or 2^-100 <= |x| <= 2^100. Keeps all divisions / square
// roots of the path inside the exponent range where the hand-expanded
// sequences (MATH_XACT) equal the IEEE ones.
static void check_range(const double *v, size_t n, const char *what) {
  for (size_t i = 0; i < n; i++) {
    const double a = std::fabs(v[i]);
    if (!(a == 0.0 || (a >= 0x1p-100 && a <= 0x1p100)))
      fail(PMAF_ERR_INVALID, std::string(what) + ": value outside the supported numeric range (finite, 0 or 2^-100 <= |x| <= 2^100)");
  }
}

struct pmaf_planner {
  DevView D{};
  int device = 0;
  int lpa = 64;
  int math = MATH_XACT;        // arithmetic policy of the w64 rollout kernels (pmaf_device.hpp)
  bool force_generic = false;  // PMAF_FORCE_GENERIC=1: always use the generic k_rollout<LPA>
  bool w64_slice = false;      // two waves of the wave-per-agent kernel on a SIMD trade issue priority (pmaf_k_w64.hip, SLICE)
  bool plain_step = false;     // every k_attr != 0 and unit mass: the wave-per-agent kernels' PLAIN step (pmaf_k_w64.hip)
  bool blocking_wait = false;  // PMAF_FLAG_BLOCKING_WAIT: pmaf_tick sleeps on an event instead of spinning on the mailbox
  bool dpp_sum = true;         // w64 kernels: ordered force sum by the DPP chain (M > 20) or LDS batches
  // W waves per agent (pmaf_k_mw.hip; 0: the wave-per-agent kernels): 61..256 field obstacles, every policy but the compiler-IEEE one
  // arithmetic, and every wave of the launch with a SIMD to itself (pick_mw)
  int mw_waves = 0, mw_per = 0, mw_lds_kb = 0;
  int n_blocks = 0;
  size_t lds_rollout = 0, lds_manager = 0;
  hipModule_t ext_mod = nullptr;       // pmaf_debug_external_rollout: a rollout kernel loaded from a code object file
  hipFunction_t ext_fn = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev_mgr = nullptr;
  uint64_t mailbox_seq = 0;     // sequence number of the last pmaf_tick (mailbox entry 11)
  double tick_timeout_s = 5.0;  // PMAF_TICK_TIMEOUT_S: bound of pmaf_tick's wait for the manager kernel's result
  bool dbg_withhold = false;    // pmaf_debug_withhold_mailbox: the sequence number is not published (fault injection)
  // winner path in mapped pinned memory (pmaf_enable_winner_path): [P][cap][3] + header [P][4] = n_points, agent, 0, seq
  double *h_wp = nullptr, *d_wp = nullptr, *h_wph = nullptr, *d_wph = nullptr;
  std::vector<int32_t> wp_np, wp_agent;
  double wp_seq = 0.0;          // sequence number the last selection published its path under (0: none yet)
  std::chrono::steady_clock::time_point last_tick_entry{};
  bool wp_timed = true;         // the last pmaf_tick's path latency has been booked (or there was no tick)
  std::vector<double> wp_us;
  // DevView::closest_idx: k_manager recomputes the table at the next reset when the caller has handed over a live
  // obstacle list that DIFFERS (bit for bit) from the previous one -- a node that passes the same static list every tick
  // pays for the table once
  bool closest_dirty = true;
  std::vector<double> last_live;   // the caller's last list, as given ([P][n_obs][7])
  // the COMPLETE list D.obs_live currently holds, as the caller gave it (empty: unknown): pmaf_tick does not hand a list
  // over again that is already resident -- the reference's node passes its obstacles_ with every call, and reading
  // 1.8 KB of mapped host memory in the manager kernel costs ~5 us of set-point latency (tools/ticklat.py)
  std::vector<double> live_resident;
  // (only the FIELD obstacles count -- the first n_obs - 1 rows of every population: the trailing repulsive obstacle,
  // which a host-coupled dual-arm run rewrites every tick, is not in the table)
  void note_live_obstacles(const double *obstacles) {
    const size_t row = (size_t)D.n_obs * 7, field = (size_t)(D.n_obs - 1) * 7;
    bool same = last_live.size() == (size_t)D.P * field;
    for (int p = 0; p < D.P && same; p++)
      same = std::memcmp(last_live.data() + p * field, obstacles + p * row, sizeof(double) * field) == 0;
    if (same) return;
    last_live.resize((size_t)D.P * field);
    for (int p = 0; p < D.P; p++) std::memcpy(last_live.data() + p * field, obstacles + p * row, sizeof(double) * field);
    closest_dirty = true;
  }
  // the kernels that read the table: the wave-per-agent kernels with several obstacle slots per lane (launch_rollout)
  // (an external kernel -- pmaf_debug_external_rollout -- is the product kernel's own code and reads the table too)
  bool uses_closest_table() const {
    const int M = D.n_obs - 1;
    const int tiles64 = (M >= 61 && M <= 64) ? 2 : (M + 63) / 64;
    return lpa == 64 && tiles64 >= 2 && tiles64 <= 4 && !force_generic;
  }
  // host-side clock of the last pmaf_tick calls (pmaf_get_tick_times_us): entry -> both launches enqueued, entry ->
  // set-point on the host; a ring of the newest TICK_RING calls
  static constexpr size_t TICK_RING = 8192;
  std::vector<float> tick_enq_us, tick_sp_us;
  size_t tick_head = 0, tick_count = 0;
  std::vector<void *> allocs;
  std::vector<size_t> alloc_bytes;  // size of every device buffer (state save / load)
  double *h_out = nullptr;      // pinned [P][PMAF_MBOX] mailbox written by k_manager
  // host copy of the real agent's state (getNextPosition / getNextVelocity /
  // getEEForce / getDistFromGoal must not wait for the running rollout)
  std::vector<double> real_pos_h, real_vel_h, real_force_h;
  double *d_out = nullptr;      // device alias of h_out
  double *h_zc = nullptr, *d_zc = nullptr;  // mapped pinned obstacle buffer read by k_manager in pmaf_tick
  // closed loop: pmaf_set_real_position leaves the measured position [P][3] in mapped pinned memory and the NEXT manager
  // launch reads it from there (ManagerArgs::real_pos_src) -- the call neither waits for the running rollout nor puts a
  // copy command in front of the tick. One buffer is enough: every entry point that launches the manager kernel returns
  // only after that kernel has read its inputs.
  double *h_rp = nullptr, *d_rp = nullptr;
  bool real_pos_pending = false;
  // pop 0's D.has_best as the host knows it (pmaf_move_real's precondition without a device round trip): 1 after any
  // completed selection, 0 on a fresh handle, -1 unknown (after pmaf_load_state / pmaf_set_best with id 0: read back once)
  int has_best_h = 0;
  double call_seq = 0.0;        // mailbox sequence numbers of the manager launches that are not pmaf_ticks: -1, -2, ...
  // pmaf_tick gave up on its time limit with its launches still queued / running (they still read h_zc / h_rp and write
  // the mailbox): no further tick until the stream has been drained (pmaf_stop)
  bool tick_abandoned = false;
  // ---- winner-record exchange of sharded runs (pmaf_attach_comm) ----
  // Two exchange slots used alternately: the selection of tick k sends from slot k & 1, so its manager kernel only has
  // to wait for the exchange of tick k-2 (long through) -- never for the collective of the tick before, which a slower
  // peer rank may not even have joined yet.
  struct Exchange {
    pmaf_comm *c = nullptr;
    hipStream_t xp = nullptr;          // pack stream: k_winner_path (local work only, never behind a collective)
    hipStream_t xs = nullptr;          // exchange stream: all-gather + copy of the table to the host
    struct Slot {
      hipEvent_t ev_pack = nullptr;    // k_winner_path has read the scored paths
      hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // around ncclAllGather (timing)
      hipEvent_t ev_done = nullptr;    // table on the host
      double *d_send = nullptr, *d_recv = nullptr;  // [P][rec], [world][P][rec]
      double *h_send = nullptr, *h_recv = nullptr;  // pinned
      bool inflight = false;
      bool pack_pending = false;       // ev_pack recorded and not yet known to have completed
    } slot[2];
    int cur = 1;                       // slot of the last exchange begun (the next one takes cur ^ 1)
    double *paths_a = nullptr, *paths_b = nullptr;  // the two path buffers (a = the handle's original one)
    bool failed = false;               // the last exchange did not complete: pmaf_winners_wait reports it
    std::vector<double> ag_us;
    bool any_inflight() const { return slot[0].inflight || slot[1].inflight; }
  } x;
  // ---- peer mailboxes (pmaf_peer_*): header-only exchange without a collective ----
  struct Peer {
    bool on = false;
    int world = 0, rank = 0;
    double *inbox = nullptr;             // this rank's inbox [2][world][P][PMAF_PEER_SLOT] (device memory)
    bool inbox_fine = false;
    std::vector<double *> mapped;        // every rank's inbox as mapped into this process
    std::vector<char> opened;            // 1: mapped[r] came from hipIpcOpenMemHandle
    PeerView *d_view = nullptr;
    int32_t *d_couple = nullptr;
    double *d_radius = nullptr;
    std::vector<int32_t> couple_h;
    std::vector<double> radius_h;
    uint64_t tick = 0;                   // pmaf_ticks since pmaf_peer_connect = sequence number last published
    double us_per_tick = 0.01;           // wall_clock64 period
    std::vector<double> wait_us, pub_us;
  } peer;
  double *d_send1 = nullptr;           // send buffer of the one-shot pmaf_allgather_winners
  // host mirror of the predicted paths (pmaf_get_paths): the reference's node calls getPredictedPaths() /
  // getNumPredictionSteps(i) 3 x N times per tick (B/src/panda_bimanual_control.cpp:341-344) -- one D2H per
  // rollout generation serves them all
  uint64_t paths_gen = 1, mirror_gen = 0;   // bumped by everything that changes paths / n_points
  double *h_paths = nullptr;                // pinned [P][N][cap][3], tails zeroed
  int32_t *h_np = nullptr;                  // pinned [P][N]
  double *d_plan_obs = nullptr;        // [P][7][n_obs] the stepping API's obstacle list (pmaf_move_agents ...)
  double *d_plan_out = nullptr;        // [P][N] pmaf_eval_obstacle_distance
  int32_t *d_plan_calls = nullptr;     // [P]
  bool stepped = false;                // agents were moved / set by the stepping API: a rollout needs a reset first
  double exchange_timeout_s = 60.0;    // PMAF_EXCHANGE_TIMEOUT_S: bound of the wait for a winner exchange
  double *d_link = nullptr, *h_link = nullptr;  // pmaf_link_force scratch (device / pinned host), grown on demand
  size_t link_scratch_doubles = 0;
  double *d_reset_in = nullptr; // [P][6]
  int32_t *d_agent_id = nullptr;// [P]
  CostParams cp{};
  bool cp_valid = false;        // cp holds the workspace terms the stored cost_ws was computed with
  bool scores_valid = false;
  bool rollout_pending = false; // agents were (re)set since the last rollout
  std::vector<std::vector<double>> real_path;  // per population, xyz triples
  std::vector<double> goal_h;
  // profiling
  bool profiling = false;
  int prof_every = 1;           // event timing on every prof_every-th rollout launch (pmaf_set_profiling's argument)
  int64_t prof_phase = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free, ev_inflight;
  double rollout_ms = 0.0, last_rollout_ms = 0.0;
  int64_t launches = 0, timed_launches = 0;

  template <typename T>
  T *dalloc(size_t n) {
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, sizeof(T) * (n ? n : 1)));
    allocs.push_back(p);
    alloc_bytes.push_back(sizeof(T) * (n ? n : 1));
    HIP_CHECK(hipMemsetAsync(p, 0, sizeof(T) * (n ? n : 1), stream));
    return static_cast<T *>(p);
  }
  // scratch that is not planner state (not in the checkpoint blob); freed with the handle
  std::vector<void *> scratch;
  template <typename T>
  T *dalloc_untracked(size_t n) {
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, sizeof(T) * (n ? n : 1)));
    scratch.push_back(p);
    return static_cast<T *>(p);
  }
  template <typename T>
  void upload(T *dst, const T *src, size_t n) {
    HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(T) * n, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  template <typename T>
  void download(T *dst, const T *src, size_t n) {
    HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(T) * n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  void use_device() { HIP_CHECK(hipSetDevice(device)); }
};

static void aos_to_soa(const double *aos, double *soa, int P, int n_obs) {
  for (int p = 0; p < P; p++)
    for (int i = 0; i < n_obs; i++)
      for (int c = 0; c < 7; c++) soa[((size_t)p * 7 + c) * n_obs + i] = aos[((size_t)p * n_obs + i) * 7 + c];
}

// smallest z >= 0 with sqrt(z) > c, i.e. (sqrt(z) > c) == (z >= sq_gt(c)) for all z >= 0.
// std::sqrt is correctly rounded on the host and bit-identical to the device's
// (tests/test_parity_gpu.py::test_device_arithmetic_is_ieee_exact).
static double sq_gt(double c) {
  double z = c * c;
  while (z > 0.0 && std::sqrt(z) > c) z = std::nextafter(z, 0.0);
  while (!(std::sqrt(z) > c)) z = std::nextafter(z, INFINITY);
  return z;
}
// smallest z >= 0 with sqrt(z) >= c, i.e. (sqrt(z) < c) == (z < sq_ge(c))
static double sq_ge(double c) {
  double z = c * c;
  while (z > 0.0 && std::sqrt(z) >= c) z = std::nextafter(z, 0.0);
  while (!(std::sqrt(z) >= c)) z = std::nextafter(z, INFINITY);
  return z;
}

// Boundary of the repulsive obstacle's range test (repelForce, B/src/cf_agent.cpp:168-171):
// smallest z >= 0 for which  max(sqrt(z) - R, 1e-5) < shell  is false; the predicate is
// monotone in z, so  (max(sqrt(z) - R, 1e-5) < shell) == (z < boundary).
static double repel_boundary(double R, double shell) {
  auto in_range = [&](double z) { double d = std::sqrt(z) - R; d = (d < 1e-5) ? 1e-5 : d; return d < shell; };
  if (!in_range(0.0)) return 0.0;
  double lo = 0.0, hi = 1.0;
  while (in_range(hi)) { hi *= 4.0; if (!(hi < 1e300)) return INFINITY; }
  // bisection on the ordered bit patterns of non-negative doubles
  uint64_t a, b;
  std::memcpy(&a, &lo, 8); std::memcpy(&b, &hi, 8);
  while (b - a > 1) {
    uint64_t m = a + (b - a) / 2;
    double z; std::memcpy(&z, &m, 8);
    if (in_range(z)) a = m; else b = m;
  }
  double z; std::memcpy(&z, &b, 8);
  return z;
}

// Which step the wave-per-agent kernels run (pmaf_k_w64.hip, PLAIN): the one without attractorForce's `k_attr != 0`
// test and the division by the mass is only valid when every agent's k_attr is non-zero and the agents have unit mass.
// Decided where the gains / the mass enter the handle: pmaf_create and pmaf_load_state (k_attr == NULL: read them back
// from the device).
static void refresh_plain_step(pmaf_planner *h, const double *k_attr) {
  const size_t PN = (size_t)h->D.P * h->D.N;
  std::vector<double> tmp;
  if (!k_attr) {
    tmp.resize(PN);
    h->download(tmp.data(), h->D.k_attr, PN);
    k_attr = tmp.data();
  }
  bool plain = (h->D.C.mass == 1.0);
  for (size_t i = 0; i < PN && plain; i++) plain = (k_attr[i] != 0.0);
  const char *ps = getenv("PMAF_PLAIN_STEP");   // "0": always the general step (tests, timing)
  if (ps && ps[0] == '0') plain = false;
  h->plain_step = plain;
}

static int pick_lpa(int N, int P, int M, int n_simds) {
  // The mapping with the smallest estimated kernel time (pmaf_lpa_model.hpp: a table of measured launch times per
  // mapping, obstacle slots per lane and waves per SIMD; profiles/r6_lpa_grid.txt). History of the rule it replaces:
  // rounds 1-4 narrowed the mapping until the launch had <= 2048 waves (the wave per agent wins while every wave has a SIMD
  // to itself and still at two per SIMD; the group kernels run best at two per SIMD); round 5 added "never more than two
  // obstacle slots per lane in a narrower mapping" (profiles/r5_lpa_rule.txt: the three- / four-slot group bodies and the
  // generic kernel cost more than another round of waves: 128 obstacles x 4096 agents 1882 -> 1149 us); round 6 measured
  // the whole plane and found that rule 20 ... 31 % off in three regions (header of pmaf_lpa_model.hpp). BASELINE's
  // configurations keep their mappings: C1-C4 the wave per agent, C5 x 8 on one GPU 16 lanes, x 4 32 lanes, x 2 / x 1 64.
  int lpa = pmaf_lpa::pick(N, P, M, n_simds);
  // known-flag bitmask holds 64 tiles per lane
  while ((M + lpa - 1) / lpa > 64 && lpa < 64) lpa *= 2;
  return lpa;
}

// W waves per agent with <= 61 obstacles each (pmaf_k_mw.hip) instead of 2 / 4 obstacle slots per lane of ONE wave:
// the per-obstacle part of the step runs on W SIMDs at once. Only while the launch leaves every BLOCK a CU of its own
// (N P <= CUs of the device) -- beyond that the multi-slot kernels' single wave per agent wins back.
// PMAF_MW=0 / 2 / 3 / 4: off / that many waves (tests, timing); PMAF_MW_PER: obstacles per wave (default: even split).
static void pick_mw(pmaf_planner *h, int N, int P, int M) {
  h->mw_waves = 0; h->mw_per = 0;
  if (h->lpa != 64 || h->force_generic || h->math == MATH_IEEE) return;
  if (M < 61 || M > 4 * 64) return;
  // as few waves as hold the obstacles at 64 per wave (every wave more costs ~0.24 us per step: profiles/r4_ab_mw.txt);
  // at <= 61 per wave lanes 61..63 stay free for the tail's riders and the sweep's norms ride along (pmaf_k_mw.hip)
  int waves = (M + 63) / 64;
  if (waves < 2) waves = 2;
  const char *e = getenv("PMAF_MW");
  if (e && e[0]) {
    const int f = atoi(e);
    if (f == 0) return;
    if (f >= waves && f <= 4) waves = f;
  }
  // ONE block per CU (pmaf_k_mw.hip's launcher enforces it through the LDS request). Rounds 4's rule let two two-wave
  // blocks share a CU (N P <= 2 CUs); measured in round 5 (profiles/r5_mw_rule_sweep.txt, 300 steps, kernel us per launch):
  //   128 obstacles: 256 agents split 536 / one-wave 578, 288 ... 512 agents split 710 ... 716 / one-wave 601
  //   100 obstacles: 256 agents 492 / 550, 384 ... 512 agents 649 ... 652 / 574;   64 obstacles: 445 / 515, 616 ... 621 / 540
  // -- as soon as ONE CU holds two blocks (their four waves contend for the CU's LDS pipe at the per-step hand-off) the
  // launch is 18 % slower than the two-slot one-wave kernel, so the split kernel is kept to launches with a CU per block.
  const long cus = h->D.n_simds / 4;
  if ((long)N * P > cus) return;
  { const char *lk = getenv("PMAF_MW_LDS_KB"); h->mw_lds_kb = lk ? atoi(lk) : 0; }   // timing experiments
  int per = (M + waves - 1) / waves;
  const char *pe = getenv("PMAF_MW_PER");
  if (pe && atoi(pe) >= per && atoi(pe) <= 64) per = atoi(pe);
  h->mw_waves = waves; h->mw_per = per;
}

// fold finished rollout event pairs (oldest first) into the stats; all=true
// requires the stream to be idle
static void drain_events(pmaf_planner *h, bool all) {
  size_t done = 0;
  for (; done < h->ev_inflight.size(); done++) {
    auto &pr = h->ev_inflight[done];
    if (!all && hipEventQuery(pr.second) != hipSuccess) break;
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
    h->rollout_ms += ms;
    h->last_rollout_ms = ms;
    h->timed_launches++;
    h->ev_free.push_back(pr);
  }
  h->ev_inflight.erase(h->ev_inflight.begin(), h->ev_inflight.begin() + (long)done);
}

// Does this handle's wave-per-agent launch run the priority-slicing loop (k_rollout_w64_sliced)? ONE predicate for the launch and
// for pmaf_get_priority_slices: more one-slot waves than SIMDs and at most two per SIMD, the kernel variants that exist with
// the loop (DPP sum, PLAIN step, strict or contracted arithmetic), and none of the routes that bypass k_rollout_w64.
static bool w64_sliced(const pmaf_planner *h) {
  const int M = h->D.n_obs - 1;
  const long waves = (long)h->D.N * h->D.P;
  return h->w64_slice && h->lpa == 64 && !h->force_generic && !h->mw_waves && !h->ext_fn && M <= 60 && h->dpp_sum && h->plain_step &&
         (h->math == MATH_XACT || h->math == MATH_FMA) && waves > h->D.n_simds && waves <= 2L * h->D.n_simds;
}

static void launch_rollout(pmaf_planner *h) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->profiling && (h->prof_phase++ % h->prof_every) == 0) {
    drain_events(h, false);
    if (h->ev_free.empty()) {
      hipEvent_t a, b;
      HIP_CHECK(hipEventCreate(&a));
      HIP_CHECK(hipEventCreate(&b));
      h->ev_free.emplace_back(a, b);
    }
    e0 = h->ev_free.back().first;
    e1 = h->ev_free.back().second;
    h->ev_free.pop_back();
    h->ev_inflight.emplace_back(e0, e1);   // both events ride on the rollout kernel's own dispatch (no marker packets)
  }
  // with a communicator attached the rollout writes the OTHER path buffer: the exchange of the selection just made
  // may still be packing the path it scored (the getters follow D.paths)
  if (h->x.c) h->D.paths = (h->D.paths == h->x.paths_a) ? h->x.paths_b : h->x.paths_a;
  // (the one-slot kernel keeps lanes 61-63 for the goal and the two speed limits and lane 60 for the repulsive obstacle:
  // 61-64 obstacles go to the split / two-slot kernels)
  const int M = h->D.n_obs - 1;
  const int tiles64 = (M >= 61 && M <= 64) ? 2 : (M + 63) / 64;
  bool ok;
  if (h->ext_fn) {
    // measurement tooling (tools/slackprof): the same launch with a kernel out of an external code object -- the
    // product kernel's own assembly with delay instructions inserted; events as marker packets around it
    void *args[] = {(void *)&h->D, (void *)&h->cp};
    if (e0) HIP_CHECK(hipEventRecord(e0, h->stream));
    HIP_CHECK(hipModuleLaunchKernel(h->ext_fn, (unsigned)h->D.N, (unsigned)h->D.P, 1, 64, 1, 1, (unsigned)h->lds_rollout,
                                    h->stream, args, nullptr));
    if (e1) HIP_CHECK(hipEventRecord(e1, h->stream));
    ok = true;
  } else if (h->mw_waves &&
             pmaf_k_launch_mw(h->D, h->cp, h->mw_waves, h->mw_per, h->math, h->plain_step, h->mw_lds_kb, h->stream, e0, e1))
    ok = true;
  else if (h->lpa == 64 && tiles64 <= 4 && !h->force_generic) {
    // (the split kernel refused the launch -- its 72 / 96 KB LDS opt-in failed on this device, or a bad PMAF_MW_LDS_KB:
    // the one-wave kernels serve every such population, bit for bit the same; pmaf_get_waves_per_agent reports 1 from now on)
    if (h->mw_waves) { (void)hipGetLastError(); h->mw_waves = 0; h->mw_per = 0; }
    // ordered force sum: the DPP chain (h->dpp_sum, see pmaf_create), LDS batches on request (pmaf_rollout_w64.hpp)
    const bool slice = w64_sliced(h);   // two one-slot waves per SIMD: the priority-slicing loop (pmaf_k_w64.hip, SLICE)
    ok = pmaf_k_launch_w64(h->D, h->cp, tiles64, h->math, h->dpp_sum, h->plain_step, h->lds_rollout, h->stream, e0, e1, slice);
  } else if (!h->force_generic && (h->lpa == 32 || h->lpa == 16 || h->lpa == 8) && (M + h->lpa - 1) / h->lpa <= 4)
    // (policy 1, the plain fast arithmetic, exists for the w64 kernels only)
    ok = pmaf_k_launch_grp(h->D, h->cp, h->lpa, (M + h->lpa - 1) / h->lpa, h->math == MATH_FAST ? MATH_XACT : h->math,
                           h->n_blocks, h->lds_rollout, h->stream, e0, e1);
  else
    ok = pmaf_k_launch_generic(h->D, h->cp, h->lpa, h->n_blocks, h->lds_rollout, h->stream, e0, e1);
  if (!ok) fail(PMAF_ERR_INVALID, "no rollout kernel for this lanes_per_agent / obstacle count in this build");
  HIP_CHECK(hipGetLastError());
  h->launches++;
  h->paths_gen++;
  h->scores_valid = true;
  h->cp_valid = true;
  h->rollout_pending = false;
}

static void sync(pmaf_planner *h) {
  HIP_CHECK(hipStreamSynchronize(h->stream));
  // an exchange in flight still reads the scored path buffer until its pack kernel is through
  for (auto &sl : h->x.slot)
    if (sl.pack_pending) { HIP_CHECK(hipEventSynchronize(sl.ev_pack)); sl.pack_pending = false; }
  if (!h->ev_inflight.empty()) drain_events(h, true);
}

static void set_cost_params(pmaf_planner *h, const double *cost_gains, const double *ws) {
  CostParams n{};
  n.k_goal_dist = cost_gains[0];
  n.k_path_len = cost_gains[1];
  n.k_safe_dist = cost_gains[2];
  n.k_workspace = cost_gains[3];
  for (int i = 0; i < 6; i++) n.ws[i] = ws[i];
  bool same_ws = h->cp_valid && n.k_workspace == h->cp.k_workspace && std::memcmp(n.ws, h->cp.ws, sizeof(n.ws)) == 0;
  h->cp = n;
  if (!same_ws) h->scores_valid = false;
  h->cp_valid = true;
}

static void ensure_scores(pmaf_planner *h) {
  if (h->scores_valid) return;
  pmaf_k_launch_score(h->D, h->cp, h->stream);
  HIP_CHECK(hipGetLastError());
  h->scores_valid = true;
}

// The caller's obstacle list (pmaf_tick, pmaf_move_real, pmaf_reset_agents): converted into the mapped pinned buffer
// k_manager reads directly. One buffer is enough: each of these calls returns only after the manager kernel that read it
// has published its result.
static const double *stage_live_obstacles_zero_copy(pmaf_planner *h, const double *obstacles) {
  if (!obstacles) return nullptr;
  const size_t n = (size_t)h->D.P * h->D.n_obs * 7;
  // the same list as the one already resident in D.obs_live (bit for bit): nothing to hand over
  // (not while peer mailboxes are connected: a coupled population's trailing row of D.obs_live is the peer's set-point then,
  // not the caller's)
  if (!h->peer.on && h->live_resident.size() == n && std::memcmp(h->live_resident.data(), obstacles, sizeof(double) * n) == 0) return nullptr;
  check_range(obstacles, n, "obstacles");
  h->note_live_obstacles(obstacles);
  h->live_resident.assign(obstacles, obstacles + n);
  aos_to_soa(obstacles, h->h_zc, h->D.P, h->D.n_obs);
  return h->d_zc;
}

// Staging marks the caller's list as resident BEFORE a manager kernel has read it. Whatever fails between the staging and
// that kernel's published result (an upload that throws, a launch error, a mailbox time-out): the list may not have reached
// D.obs_live, so it must not count as resident -- a retry with the same list has to hand it over again. Armed on
// construction; the caller disarms it once the mailbox result of the launch that carried the list is in.
struct ResidentGuard {
  pmaf_planner *h; bool armed = true;
  ~ResidentGuard() { if (armed) { h->live_resident.clear(); h->last_live.clear(); h->closest_dirty = true; } }
};

// a pending closed-loop position into D.real_pos the slow way (stream sync + copy), for the consumers of D.real_pos
// that are not manager launches (winner-record packing, the state blob)
static void flush_real_position(pmaf_planner *h) {
  if (!h->real_pos_pending) return;
  sync(h);
  h->upload(h->D.real_pos, h->real_pos_h.data(), (size_t)h->D.P * 3);
  h->real_pos_pending = false;
}

static void launch_manager(pmaf_planner *h, const ManagerArgs &A0, hipEvent_t done = nullptr) {
  ManagerArgs A = A0;
  const bool hand_over_position = h->real_pos_pending;
  if (hand_over_position) {
    if (h->D.P <= PMAF_RP_INLINE) {   // by value in the kernel arguments (no PCIe read in front of the real step)
      A.real_pos_inline = 1;
      std::memcpy(A.real_pos_val, h->h_rp, sizeof(double) * 3 * (size_t)h->D.P);
    } else {
      A.real_pos_src = h->d_rp;
    }
  }
  if (A.do_reset) h->paths_gen++;
  if (A.do_reset && h->uses_closest_table()) { A.compute_closest = h->closest_dirty ? 1 : 0; h->closest_dirty = false; }
  A.tuned_real_step = (h->math == MATH_XACT && !h->force_generic) ? 1 : 0;
  pmaf_k_launch_manager(h->D, h->cp, A, h->lds_manager, h->stream, done);
  HIP_CHECK(hipGetLastError());
  if (hand_over_position) h->real_pos_pending = false;   // (only once the launch that reads it is in the stream)
}

// copy the mailbox into the host-side real-agent state (call after the
// k_manager launch has completed)
static void refresh_real_cache(pmaf_planner *h) {
  for (int p = 0; p < h->D.P; p++) {
    const double *o = h->h_out + p * PMAF_MBOX;
    for (int c = 0; c < 3; c++) {
      h->real_pos_h[p * 3 + c] = o[1 + c];
      h->real_vel_h[p * 3 + c] = o[4 + c];
      h->real_force_h[p * 3 + c] = o[8 + c];
    }
  }
}

// Spin until k_manager has published sequence number `seq` for every
// population. The stream is queried now and then so that a failed or finished
// launch cannot leave the host spinning.
// Bounded in wall-clock time (round 4): a hung or wedged GPU must not hang a 100 Hz control loop at 100 % CPU -- after
// PMAF_TICK_TIMEOUT_S (default 5 s) the call fails with PMAF_ERR_DEVICE (the clock is only read every 2^14 spins / on
// every poll of the blocking variant).
static void wait_mailbox(pmaf_planner *h, double seq, const char *who = "pmaf_tick") {
  std::chrono::steady_clock::time_point t_start{};
  bool timing = false;
  for (int p = 0; p < h->D.P; p++) {
    const volatile double *s = h->h_out + p * PMAF_MBOX + 11;
    unsigned spins = 0;
    while (*s != seq) {
      if (h->blocking_wait) {
        // PMAF_FLAG_BLOCKING_WAIT: give the core away between polls (batch drivers that issue ticks back to back
        // would otherwise spin for the rest of the previous rollout); costs up to one sleep quantum of latency
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        spins += 0x3ff;
      } else {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
      }
      if ((++spins & 0x3fffu) == 0 || (h->blocking_wait && (spins & 0x3fffu) < 0x400)) {
        const auto now = std::chrono::steady_clock::now();
        if (!timing) { timing = true; t_start = now; }
        else if (std::chrono::duration<double>(now - t_start).count() > h->tick_timeout_s) {
          // the two launches of this tick are still queued or running: they will read the zero-copy buffers and write
          // the mailbox later, so the handle takes no further tick until the stream has been drained (pmaf_stop)
          h->tick_abandoned = true;
          char msg[384];
          snprintf(msg, sizeof msg, "%s: no result from the manager kernel within the time limit (PMAF_TICK_TIMEOUT_S = %g s): "
                   "device hung, the previous rollout still running%s; call pmaf_stop() before the next tick", who, h->tick_timeout_s,
                   h->peer.on ? ", or a coupled peer's header outstanding (the in-kernel wait is bounded by PMAF_PEER_TIMEOUT_S)" : "");
          fail(PMAF_ERR_DEVICE, msg);
        }
        hipError_t e = hipStreamQuery(h->stream);
        if (e == hipErrorNotReady) continue;
        if (e != hipSuccess) throw HipError{e, "hipStreamQuery (mailbox wait)", __LINE__};
        if (*s != seq) fail(PMAF_ERR_DEVICE, std::string(who) + ": manager kernel finished without publishing its result");
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

static void append_real_path(pmaf_planner *h) {
  for (int p = 0; p < h->D.P; p++) {
    const double *o = h->h_out + p * PMAF_MBOX;
    h->real_path[p].insert(h->real_path[p].end(), {o[1], o[2], o[3]});
  }
}

// ---- winner-record exchange ---------------------------------------------------
static size_t winner_rec(const pmaf_planner *h) { return PMAF_WINNER_HDR + (size_t)h->D.cap * 3; }

// complete the exchange in flight (if any): RCCL -- wait for the table on the host and book the all-gather's device
// time; host transport -- wait for the packed records, then run the caller's collective here
static void finish_exchange(pmaf_planner *h, int which) {
  pmaf_planner::Exchange &x = h->x;
  pmaf_planner::Exchange::Slot &sl = x.slot[which];
  if (!sl.inflight) return;
  // Whatever happens below, this exchange is over: a failed one must not be retried by the next call (the ranks
  // would no longer issue their collectives in the same order)
  sl.inflight = false;
  {
    // normally long through; poll instead of a blocking wait (whose wake-up latency would land on the next tick's
    // enqueue when the exchange is still in flight). Bounded: a peer that died leaves ncclAllGather waiting for ever --
    // after PMAF_EXCHANGE_TIMEOUT_S (default 60 s) the call fails with PMAF_ERR_DEVICE instead of spinning on
    hipError_t e;
    unsigned spins = 0;
    std::chrono::steady_clock::time_point t_start{};
    bool timing = false;
    while ((e = hipEventQuery(sl.ev_done)) == hipErrorNotReady) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#endif
      if ((++spins & 0xfffffu) == 0) {
        std::this_thread::yield();
        const auto now = std::chrono::steady_clock::now();
        if (!timing) { timing = true; t_start = now; }
        else if (std::chrono::duration<double>(now - t_start).count() > h->exchange_timeout_s) {
          x.failed = true;
          fail(PMAF_ERR_DEVICE, "winner exchange: the all-gather did not complete within the time limit (a peer rank "
                                "gone? PMAF_EXCHANGE_TIMEOUT_S)");
        }
      }
    }
    if (e != hipSuccess) { x.failed = true; throw HipError{e, "hipEventQuery (winner exchange)", __LINE__}; }
  }
  sl.pack_pending = false;
  const size_t n_local = (size_t)h->D.P * winner_rec(h);
  if (x.c->rccl) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, sl.ev_t0, sl.ev_t1));
    x.ag_us.push_back((double)ms * 1e3);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    if (x.c->fn(x.c->ctx, sl.h_send, sl.h_recv, n_local * sizeof(double)) != 0) {
      x.failed = true;
      fail(PMAF_ERR_DEVICE, "winner exchange: the host all-gather callback failed");
    }
    x.ag_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    HIP_CHECK(hipMemcpyAsync(sl.d_recv, sl.h_recv, n_local * x.c->world * sizeof(double), hipMemcpyHostToDevice, x.xs));
    HIP_CHECK(hipStreamSynchronize(x.xs));
  }
  if (x.ag_us.size() > (1u << 20)) x.ag_us.erase(x.ag_us.begin(), x.ag_us.begin() + (1u << 19));
}
// all exchanges in flight, oldest first (the host transport runs the callers' collectives here: same order on every rank)
static void finish_exchanges(pmaf_planner *h) {
  finish_exchange(h, h->x.cur ^ 1);
  finish_exchange(h, h->x.cur);
}
// the slot the next selection will send from: its previous exchange (two selections ago) must be through, and so
// must the pack kernel of the LAST exchange, which reads the path buffer the rollout launched next overwrites
// (local work on its own stream -- this wait never involves a peer)
static pmaf_planner::Exchange::Slot &claim_exchange_slot(pmaf_planner *h) {
  pmaf_planner::Exchange &x = h->x;
  finish_exchange(h, x.cur ^ 1);
  pmaf_planner::Exchange::Slot &last = x.slot[x.cur];
  if (last.pack_pending) {
    if (hipEventQuery(last.ev_pack) != hipSuccess) HIP_CHECK(hipEventSynchronize(last.ev_pack));
    last.pack_pending = false;
  }
  return x.slot[x.cur ^ 1];
}

// pack the selected agents' paths out of `scored_paths` behind the headers k_manager wrote into the claimed slot's
// send buffer and all-gather the records.
// Called once the HOST knows that k_manager has written the record headers (pmaf_tick: the mailbox's sequence number
// has arrived, and the headers are stored before it behind a system-scope fence; pmaf_evaluate: the stream is
// synchronised): the exchange streams then need no event from the handle's stream -- a cross-stream event between
// the manager and the rollout cost 4-7 us per tick (measured), this costs the rollout nothing.
static void begin_exchange(pmaf_planner *h, const double *scored_paths) {
  pmaf_planner::Exchange &x = h->x;
  x.cur ^= 1;
  pmaf_planner::Exchange::Slot &sl = x.slot[x.cur];
  const size_t n_local = (size_t)h->D.P * winner_rec(h);
  pmaf_k_launch_winner_path(h->D, scored_paths, sl.d_send, x.xp);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventRecord(sl.ev_pack, x.xp));
  sl.pack_pending = true;
  HIP_CHECK(hipStreamWaitEvent(x.xs, sl.ev_pack, 0));
  if (x.c->rccl) {
    HIP_CHECK(hipEventRecord(sl.ev_t0, x.xs));
    const std::string err = pmaf_comm_enqueue_allgather(x.c, sl.d_send, sl.d_recv, n_local, x.xs);
    if (!err.empty()) fail(PMAF_ERR_DEVICE, err);
    HIP_CHECK(hipEventRecord(sl.ev_t1, x.xs));
    HIP_CHECK(hipMemcpyAsync(sl.h_recv, sl.d_recv, n_local * x.c->world * sizeof(double), hipMemcpyDeviceToHost, x.xs));
  } else {
    HIP_CHECK(hipMemcpyAsync(sl.h_send, sl.d_send, n_local * sizeof(double), hipMemcpyDeviceToHost, x.xs));
  }
  HIP_CHECK(hipEventRecord(sl.ev_done, x.xs));
  sl.inflight = true;
  x.failed = false;
}

static void detach_comm(pmaf_planner *h) {
  pmaf_planner::Exchange &x = h->x;
  if (!x.c) return;
  h->paths_gen++;
  // an RCCL all-gather already enqueued completes (every rank enqueued it); a host collective not yet run is dropped
  if (x.xp) (void)hipStreamSynchronize(x.xp);
  if (x.any_inflight() && x.xs) (void)hipStreamSynchronize(x.xs);
  (void)hipStreamSynchronize(h->stream);
  if (h->D.paths != x.paths_a) {  // the handle goes back to its own path buffer
    (void)hipMemcpy(x.paths_a, x.paths_b, sizeof(double) * (size_t)h->D.P * h->D.N * h->D.cap * 3, hipMemcpyDeviceToDevice);
    h->D.paths = x.paths_a;
  }
  if (x.paths_b) (void)hipFree(x.paths_b);
  for (auto &sl : x.slot) {
    if (sl.d_send) (void)hipFree(sl.d_send);
    if (sl.d_recv) (void)hipFree(sl.d_recv);
    if (sl.h_send) (void)hipHostFree(sl.h_send);
    if (sl.h_recv) (void)hipHostFree(sl.h_recv);
    for (hipEvent_t e : {sl.ev_pack, sl.ev_t0, sl.ev_t1, sl.ev_done}) if (e) (void)hipEventDestroy(e);
  }
  if (x.xp) (void)hipStreamDestroy(x.xp);
  if (x.xs) (void)hipStreamDestroy(x.xs);
  x = pmaf_planner::Exchange{};
}

// ---- peer mailboxes ----------------------------------------------------------
struct PeerHandleBlob {   // what pmaf_peer_export hands out (PMAF_PEER_HANDLE_BYTES)
  uint64_t magic;
  int64_t pid;
  uint64_t ptr;           // the inbox in the exporting process (used when the importer IS that process)
  int32_t world, P, device, fine;   // fine: the inbox is fine-grained device memory
  hipIpcMemHandle_t ipc;
  int32_t pci[3];                   // PCI domain / bus / device of the exporting GPU (ordinals differ between processes)
  unsigned char fill[PMAF_PEER_HANDLE_BYTES - 40 - sizeof(hipIpcMemHandle_t) - 12];
};
static void pci_id_of(int device, int32_t out[3]) {
  int v = 0;
  out[0] = (hipDeviceGetAttribute(&v, hipDeviceAttributePciDomainID, device) == hipSuccess) ? v : -1;
  out[1] = (hipDeviceGetAttribute(&v, hipDeviceAttributePciBusId, device) == hipSuccess) ? v : -1;
  out[2] = (hipDeviceGetAttribute(&v, hipDeviceAttributePciDeviceId, device) == hipSuccess) ? v : -1;
  (void)hipGetLastError();
}
static_assert(sizeof(PeerHandleBlob) == PMAF_PEER_HANDLE_BYTES, "pmaf.h: PMAF_PEER_HANDLE_BYTES");
static const uint64_t kPeerMagic = 0x504d41465f505232ull;  // "PMAF_PR2" (the blob of ABI 5 changed layout: pad -> fine, pci[3])

static size_t peer_inbox_doubles(int world, int P) { return (size_t)2 * world * P * PMAF_PEER_SLOT; }

// after the mailbox of a tick has arrived: the wait for the coupled header / the publish time of this tick's manager
// kernel, and whether a header it waited for never came
// returns the worst peer status of the tick: 0 ok, 1 time-out, 2 the source ran ahead, 3 coupling not mutual
static int peer_book_tick(pmaf_planner *h) {
  pmaf_planner::Peer &pr = h->peer;
  double w = 0.0, pb = 0.0;
  int late = 0;
  for (int p = 0; p < h->D.P; p++) {
    const double *o = h->h_out + p * PMAF_MBOX;
    w = std::max(w, o[12]);
    pb = std::max(pb, o[13]);
    late = std::max(late, (int)o[14]);
  }
  pr.wait_us.push_back(w * pr.us_per_tick);
  pr.pub_us.push_back(pb * pr.us_per_tick);
  if (pr.wait_us.size() > (1u << 20)) {
    pr.wait_us.erase(pr.wait_us.begin(), pr.wait_us.begin() + (1u << 19));
    pr.pub_us.erase(pr.pub_us.begin(), pr.pub_us.begin() + (1u << 19));
  }
  return late;
}

static void peer_disconnect(pmaf_planner *h) {
  pmaf_planner::Peer &pr = h->peer;
  h->live_resident.clear();   // (a coupled trailing obstacle was written on the device: the caller's list is handed over again)
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (size_t r = 0; r < pr.mapped.size(); r++)
    if (pr.opened[r] && pr.mapped[r]) (void)hipIpcCloseMemHandle(pr.mapped[r]);
  if (pr.d_view) (void)hipFree(pr.d_view);
  if (pr.d_couple) (void)hipFree(pr.d_couple);
  if (pr.d_radius) (void)hipFree(pr.d_radius);
  if (pr.inbox) (void)hipFree(pr.inbox);
  pr = pmaf_planner::Peer{};
}

template <typename F>
static int guarded(F &&f) {
  try {
    f();
    return PMAF_OK;
  } catch (const StatusError &e) {
    g_err = e.msg;
    return e.code;
  } catch (const HipError &e) {
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s [pmaf_host.cpp:%d]", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
    g_err = buf;
    return PMAF_ERR_DEVICE;
  } catch (const std::bad_alloc &) {
    g_err = "host allocation failed";
    return PMAF_ERR_NOMEM;
  } catch (...) {
    g_err = "unknown error";
    return PMAF_ERR_INVALID;
  }
}

#define REQUIRE(c, msg) do { if (!(c)) fail(PMAF_ERR_INVALID, msg); } while (0)


extern "C" {

const char *pmaf_last_error(void) { return g_err.c_str(); }
int pmaf_abi_version(void) { return PMAF_ABI_VERSION; }
int pmaf_eval_order(void) {
#ifdef PMAF_DOT_RIGHT_ASSOC
  return PMAF_EVAL_ORDER_DOT_RIGHT;
#else
  return PMAF_EVAL_ORDER_DOT_LEFT;
#endif
}

int pmaf_create(const pmaf_params *prm, pmaf_planner **out) {
  pmaf_planner *h = nullptr;
  int rc = guarded([&] {
    REQUIRE(prm && out, "pmaf_create: NULL argument");
    REQUIRE(prm->abi_version == PMAF_ABI_VERSION, "pmaf_create: ABI version mismatch");
    REQUIRE(prm->n_populations >= 1 && prm->n_agents >= 1, "pmaf_create: need n_populations >= 1 and n_agents >= 1");
    REQUIRE(prm->n_obstacles >= 1, "pmaf_create: obstacle list must hold at least the trailing repulsive obstacle");
    REQUIRE(prm->max_prediction_steps >= 1, "pmaf_create: max_prediction_steps must be >= 1");
    REQUIRE(prm->goal && prm->obstacles && prm->k_attr && prm->k_circ && prm->k_repel && prm->k_damp,
            "pmaf_create: goal, obstacles and gain arrays are required");
    {
      const size_t P_ = (size_t)prm->n_populations, N_ = (size_t)prm->n_agents, O_ = (size_t)prm->n_obstacles;
      check_range(prm->goal, P_ * 3, "pmaf_create: goal");
      if (prm->init_pos) check_range(prm->init_pos, P_ * 3, "pmaf_create: init_pos");
      check_range(prm->obstacles, P_ * O_ * 7, "pmaf_create: obstacles");
      check_range(prm->k_attr, P_ * N_, "pmaf_create: k_attr"); check_range(prm->k_circ, P_ * N_, "pmaf_create: k_circ");
      check_range(prm->k_repel, P_ * N_, "pmaf_create: k_repel"); check_range(prm->k_damp, P_ * N_, "pmaf_create: k_damp");
      if (prm->random_vecs) check_range(prm->random_vecs, P_ * N_ * O_ * 3, "pmaf_create: random_vecs");
      const double sc[6] = {prm->dt, prm->velocity_max, prm->approach_dist, prm->detect_shell_rad, prm->agent_mass, prm->radius};
      check_range(sc, 6, "pmaf_create: scalar parameter");
    }
    int lp = prm->lanes_per_agent;
    REQUIRE(lp == 0 || (lp >= 1 && lp <= 64 && (lp & (lp - 1)) == 0), "pmaf_create: lanes_per_agent must be 0 or a power of two <= 64");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) fail(PMAF_ERR_DEVICE, "pmaf_create: no HIP device available (this library has no CPU fallback)");
    h = new pmaf_planner();
    if (prm->device >= 0) h->device = prm->device; else HIP_CHECK(hipGetDevice(&h->device));
    REQUIRE(h->device < ndev, "pmaf_create: device ordinal out of range");
    h->use_device();
    HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&h->ev_mgr, hipEventDisableTiming));
    const int P = prm->n_populations, N = prm->n_agents, n_obs = prm->n_obstacles, cap = prm->max_prediction_steps;
    const int M = n_obs - 1;
    DevView &D = h->D;
    D.P = P; D.N = N; D.n_obs = n_obs; D.cap = cap;
    D.C.dt = prm->dt; D.C.vel_max = prm->velocity_max; D.C.approach = prm->approach_dist;
    D.C.shell = prm->detect_shell_rad; D.C.mass = prm->agent_mass; D.C.rad = prm->radius;
    D.C.zf_gt = sq_gt(1e-5); D.C.zacc_gt = sq_gt(13.0); D.C.zinit_lt = sq_ge(0.2);
    D.C.zvhalf_lt = sq_ge(0.5 * prm->velocity_max);
    D.C.zv09_lt = sq_ge(prm->velocity_max - 0.1 * prm->velocity_max);
    h->math = (prm->flags & PMAF_FLAG_CONTRACTED) ? MATH_FMA : (prm->flags & PMAF_FLAG_FAST_MATH) ? MATH_FAST
              : (prm->flags & PMAF_FLAG_IEEE_SEQUENCES) ? MATH_IEEE : MATH_XACT;
    h->blocking_wait = (prm->flags & PMAF_FLAG_BLOCKING_WAIT) != 0;
    {
      int cus = 0;
      HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
      D.n_simds = 4 * (cus > 0 ? cus : 256);
    }
    h->lpa = lp ? lp : pick_lpa(N, P, M, D.n_simds);
    { const char *fg = getenv("PMAF_FORCE_GENERIC"); h->force_generic = fg && fg[0] == '1'; }
    // Two waves of the wave-per-agent kernel on one SIMD (1 025 ... 2 048 agents in the handle) trade issue priority in slices
    // of the wall clock so that both finish together (pmaf_k_w64.hip, SLICE; pmaf_get_priority_slices). tools/slicesweep.py,
    // profiles/r6_slice_sweep.txt, kernel us per launch without -> with (slices of 2^9 ticks = 5.1 us, the younger wave 5 of 8):
    //   32 obstacles: 1 280 agents 347 -> 326, 1 536: 359 -> 334, 2 048: 381 -> 364;  9 x 2 048: 376 -> 362;  60 x 2 048: 383 -> 370;
    //   BASELINE C5, two scenes in the handle (its per-GPU load at 4 GPUs): 386 -> 372.  Settings 2^8 ... 2^10 x 4 ... 6 of 8: within 2 %.
    // The arithmetic and its order are the same instructions: bit-identical results (tests/test_parity_gpu.py runs these shapes).
    // PMAF_W64_SLICE=0|1, PMAF_W64_SLICE_LOG2, PMAF_W64_SLICE_YOUNGER in the environment: timing experiments.
    { const char *e = getenv("PMAF_W64_SLICE"); h->w64_slice = e ? (e[0] == '1') : PMAF_W64_SLICE_DEFAULT; }
    { const char *e = getenv("PMAF_W64_SLICE_LOG2"); D.prio_slice_log2 = std::min(20, std::max(4, e ? atoi(e) : 9)); }       // (a shift count on the device)
    { const char *e = getenv("PMAF_W64_SLICE_YOUNGER"); D.prio_younger_of_8 = std::min(7, std::max(1, e ? atoi(e) : 5)); }
    { const char *ab = getenv("PMAF_ABLATE"); D.ablate = ab ? atoi(ab) : 0; }
    { const char *to = getenv("PMAF_EXCHANGE_TIMEOUT_S"); if (to && atof(to) > 0.0) h->exchange_timeout_s = atof(to); }
    { const char *to = getenv("PMAF_TICK_TIMEOUT_S"); if (to && atof(to) > 0.0) h->tick_timeout_s = atof(to); }
    // ordered force sum: the DPP chain for every obstacle count (round 3: with the first chunk's accumulates fused and
    // interleaved with the scaling chain it also wins for short lists -- C1, nine obstacles: 121.3 -> 111.8 us; rounds 1-2
    // switched to LDS batches below 21 obstacles). PMAF_SUM=lds selects the LDS-batch kernels (tests, timing).
    h->dpp_sum = true;
    { const char *ds = getenv("PMAF_SUM"); if (ds && ds[0]) h->dpp_sum = (ds[0] == 'd'); }  // "dpp" / "lds": tests, timing
    pick_mw(h, N, P, M);
    REQUIRE((M + h->lpa - 1) / h->lpa <= 64, "pmaf_create: too many obstacles for this lanes_per_agent (need M <= 64*lanes_per_agent)");
    h->n_blocks = (N * h->lpa + 63) / 64;
    {
      // obstacle table + known flags, then (w64 kernels) the per-step list of
      // circular-field terms: 64 * TILES entries of 4 doubles
      size_t off = 7 * (size_t)n_obs + ((size_t)n_obs + 1) / 2;
      off += off & 1;
      // w64: (64 * TILES + 8 padding + 64 scratch) entries; groups: 64 * TILES + one zero entry per group (<= 8).
      // TILES as dispatched (launch_rollout), never below 2: with the four-slot size a one-wave block of 129 obstacles
      // asks for 18.5 KB + the kernel's 2.1 KB of static LDS (exp's table) -- over the 20 KB that let eight blocks
      // share a CU, and 2048+ agents x 128 obstacles ran at 7/8 occupancy with a second round of blocks (+45 %).
      const int slots = (M + h->lpa - 1) / h->lpa;
      const bool narrow = h->lpa == 64 || h->lpa == 32 || h->lpa == 16 || h->lpa == 8;
      const int tiles = (narrow && !h->force_generic && slots <= 2) ? 2 : 4;
      h->lds_rollout = sizeof(double) * (off + (size_t)pmaf_list_area_doubles(tiles) + 8 * 4);
    }
    // table | known flags | costs | the tuned real step's list (64 * 4 + 8 + 64 entries of 4 doubles)
    // (+ 2: the wave-minimum cell of circ_and_scale_w64 behind the list)
    h->lds_manager = sizeof(double) * (7 * (size_t)n_obs + ((size_t)n_obs + 1) / 2 + (size_t)N + 1 + (size_t)pmaf_list_area_doubles(4) + 2);
    REQUIRE(h->lds_rollout <= 160 * 1024, "pmaf_create: obstacle table does not fit in LDS");
    REQUIRE(h->lds_manager <= 160 * 1024,
            "pmaf_create: n_agents + obstacle table exceed the manager kernel's LDS budget (160 KB: 8 B per agent, 60 B per obstacle)");
    // dynamic LDS beyond the 64 KB default needs the per-function opt-in
    HIP_CHECK(pmaf_k_set_lds_limits(h->lds_manager, h->lds_rollout));

    size_t PN = (size_t)P * N;
    double *goal = h->dalloc<double>(P * 3);
    D.goal = goal;
    D.agent_init_pos = h->dalloc<double>(P * 3);
    D.start_pos = h->dalloc<double>(P * 3);
    D.start_vel = h->dalloc<double>(P * 3);
    D.obs_start = h->dalloc<double>((size_t)P * 7 * n_obs);
    D.known_start = h->dalloc<int32_t>((size_t)P * n_obs);
    D.obs_live = h->dalloc<double>((size_t)P * 7 * n_obs);
    D.closest_idx = h->dalloc<int32_t>((size_t)P * n_obs);   // (zero-filled; closest_ok = 0: no table yet)
    D.closest_ok = h->dalloc<int32_t>((size_t)P);
    double *ka = h->dalloc<double>(PN), *kc = h->dalloc<double>(PN), *kr = h->dalloc<double>(PN), *kd = h->dalloc<double>(PN);
    D.k_attr = ka; D.k_circ = kc; D.k_repel = kr; D.k_damp = kd;
    int32_t *types = h->dalloc<int32_t>(N);
    D.types = types;
    D.rot = h->dalloc<double>(PN * 3 * n_obs);
    double *rnd = h->dalloc<double>(PN * 3 * n_obs);
    D.rnd = rnd;
    D.paths = h->dalloc<double>(PN * (size_t)cap * 3);
    D.n_points = h->dalloc<int32_t>(PN);
    D.agent_vel = h->dalloc<double>(PN * 3);
    D.min_obs = h->dalloc<double>(PN);
    D.cost_ws = h->dalloc<double>(PN);
    D.path_len = h->dalloc<double>(PN);
    D.goal_dist = h->dalloc<double>(PN);
    D.reached = h->dalloc<int32_t>(PN);
    D.known_out = h->dalloc<int32_t>(PN * n_obs);
    D.costs = h->dalloc<double>(PN);
    D.real_pos = h->dalloc<double>(P * 3);
    D.real_vel = h->dalloc<double>(P * 3);
    D.real_force = h->dalloc<double>(P * 3);
    D.real_init_pos = h->dalloc<double>(P * 3);
    D.real_known = h->dalloc<int32_t>((size_t)P * n_obs);
    D.real_rot = h->dalloc<double>((size_t)P * 3 * n_obs);
    D.has_best = h->dalloc<int32_t>(P);
    D.best_id = h->dalloc<int32_t>(P);
    D.best_type = h->dalloc<int32_t>(P);
    D.best_rnd = h->dalloc<double>((size_t)P * 3 * n_obs);
    D.best_idx = h->dalloc<int32_t>(P);
    D.step_counter = h->dalloc<unsigned long long>(1);
    D.pred_ticks = h->dalloc<unsigned long long>(PN);
    double *zsent = h->dalloc<double>(P);
    D.zsent_lt = zsent;
    h->d_reset_in = h->dalloc<double>(P * 6);
    h->d_agent_id = h->dalloc<int32_t>(P);
    HIP_CHECK(hipHostMalloc((void **)&h->h_out, sizeof(double) * P * PMAF_MBOX, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h->d_out, h->h_out, 0));
    HIP_CHECK(hipHostMalloc((void **)&h->h_zc, sizeof(double) * (size_t)P * 7 * n_obs, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h->d_zc, h->h_zc, 0));
    HIP_CHECK(hipHostMalloc((void **)&h->h_rp, sizeof(double) * (size_t)P * 3, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h->d_rp, h->h_rp, 0));
    std::memset(h->h_out, 0, sizeof(double) * P * PMAF_MBOX);

    // ---- initial state = freshly constructed agents (cf_agent.h:69-97) ----
    std::vector<double> init(P * 3, 0.0);
    if (prm->init_pos) init.assign(prm->init_pos, prm->init_pos + P * 3);
    h->goal_h.assign(prm->goal, prm->goal + P * 3);
    h->upload(goal, prm->goal, P * 3);
    // CfAgent::init_pos_ member starts at zero (cf_agent.h:78), positions at CfManager::init_pos_
    h->upload(D.start_pos, init.data(), P * 3);
    h->upload(D.real_pos, init.data(), P * 3);
    std::vector<double> v0(P * 3, 0.0);
    for (int p = 0; p < P; p++) v0[p * 3] = 0.01;  // vel_{0.01, 0, 0}, cf_agent.h:76
    h->upload(D.start_vel, v0.data(), P * 3);
    h->upload(D.real_vel, v0.data(), P * 3);
    std::vector<double> soa((size_t)P * 7 * n_obs);
    aos_to_soa(prm->obstacles, soa.data(), P, n_obs);
    {
      std::vector<double> zs(P);
      for (int p = 0; p < P; p++)
        zs[p] = repel_boundary(prm->radius + prm->obstacles[((size_t)p * n_obs + (n_obs - 1)) * 7 + 6], prm->detect_shell_rad);
      h->upload(zsent, zs.data(), P);
    }
    h->upload(D.obs_start, soa.data(), soa.size());
    h->upload(D.obs_live, soa.data(), soa.size());
    h->live_resident.assign(prm->obstacles, prm->obstacles + (size_t)P * n_obs * 7);
    h->upload(ka, prm->k_attr, PN); h->upload(kc, prm->k_circ, PN);
    refresh_plain_step(h, prm->k_attr);
    h->upload(kr, prm->k_repel, PN); h->upload(kd, prm->k_damp, PN);
    std::vector<int32_t> ty(N);
    static const int layout[5] = {PMAF_HAD_HEURISTIC, PMAF_GOAL_HEURISTIC, PMAF_OBSTACLE_HEURISTIC,
                                  PMAF_GOAL_OBSTACLE_HEURISTIC, PMAF_VEL_HEURISTIC};
    for (int i = 0; i < N; i++) {
      ty[i] = prm->agent_types ? prm->agent_types[i] : (i < 5 ? layout[i] : PMAF_RANDOM_AGENT);
      REQUIRE(ty[i] >= PMAF_GOAL_HEURISTIC && ty[i] <= PMAF_HAD_HEURISTIC, "pmaf_create: agent type must be one of the six heuristics");
    }
    h->upload(types, ty.data(), N);
    // rotation vectors start at (0,0,1), cf_agent.h:92-96; component-major [3][n_obs]
    {
      std::vector<double> rot(PN * 3 * n_obs, 0.0);
      for (size_t pa = 0; pa < PN; pa++)
        for (int i = 0; i < n_obs; i++) rot[(pa * 3 + 2) * n_obs + i] = 1.0;
      h->upload(D.rot, rot.data(), rot.size());
      std::vector<double> rr((size_t)P * 3 * n_obs, 0.0);
      for (int p = 0; p < P; p++)
        for (int i = 0; i < n_obs; i++) rr[((size_t)p * 3 + 2) * n_obs + i] = 1.0;
      h->upload(D.real_rot, rr.data(), rr.size());
    }
    if (prm->random_vecs) {
      std::vector<double> r(PN * 3 * n_obs);
      for (size_t pa = 0; pa < PN; pa++)
        for (int i = 0; i < n_obs; i++)
          for (int c = 0; c < 3; c++) r[(pa * 3 + c) * n_obs + i] = prm->random_vecs[(pa * n_obs + i) * 3 + c];
      h->upload(rnd, r.data(), r.size());
    }
    // 1-point paths at init_pos, min_obs = shell
    {
      std::vector<double> paths(PN * (size_t)cap * 3, 0.0);
      std::vector<int32_t> np(PN, 1);
      std::vector<double> mo(PN, prm->detect_shell_rad), av(PN * 3, 0.0);
      for (size_t pa = 0; pa < PN; pa++) {
        int p = (int)(pa / N);
        for (int c = 0; c < 3; c++) paths[pa * cap * 3 + c] = init[p * 3 + c];
        av[pa * 3] = 0.01;
      }
      h->upload(D.paths, paths.data(), paths.size());
      h->upload(D.n_points, np.data(), np.size());
      h->upload(D.min_obs, mo.data(), mo.size());
      h->upload(D.agent_vel, av.data(), av.size());
    }
    h->real_pos_h = init;
    h->real_vel_h = v0;
    h->real_force_h.assign(P * 3, 0.0);
    h->real_path.assign(P, {});
    for (int p = 0; p < P; p++) h->real_path[p].insert(h->real_path[p].end(), {init[p * 3], init[p * 3 + 1], init[p * 3 + 2]});
    h->rollout_pending = true;
    sync(h);
    *out = h;
  });
  if (rc != PMAF_OK && h) { pmaf_destroy(h); }
  return rc;
}

int pmaf_destroy(pmaf_planner *h) {
  if (!h) return PMAF_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  detach_comm(h);
  peer_disconnect(h);
  if (h->d_send1) (void)hipFree(h->d_send1);
  if (h->ext_mod) (void)hipModuleUnload(h->ext_mod);
  for (void *p : h->allocs) (void)hipFree(p);
  for (void *p : h->scratch) (void)hipFree(p);
  if (h->h_out) (void)hipHostFree(h->h_out);
  if (h->h_zc) (void)hipHostFree(h->h_zc);
  if (h->h_rp) (void)hipHostFree(h->h_rp);
  if (h->h_wp) (void)hipHostFree(h->h_wp);
  if (h->h_wph) (void)hipHostFree(h->h_wph);
  if (h->h_paths) (void)hipHostFree(h->h_paths);
  if (h->h_np) (void)hipHostFree(h->h_np);
  if (h->d_link) (void)hipFree(h->d_link);
  if (h->h_link) (void)hipHostFree(h->h_link);
  for (auto &e : h->ev_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto &e : h->ev_inflight) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (h->ev_mgr) (void)hipEventDestroy(h->ev_mgr);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PMAF_OK;
}

int pmaf_set_initial_position(pmaf_planner *h, const double *pos) {
  return guarded([&] {
    REQUIRE(h && pos, "pmaf_set_initial_position: NULL argument");
    check_range(pos, (size_t)h->D.P * 3, "pmaf_set_initial_position");
    h->use_device();
    sync(h);
    DevView &D = h->D;
    const int P = D.P;
    h->upload(D.agent_init_pos, pos, P * 3);
    h->upload(D.real_init_pos, pos, P * 3);
    h->upload(D.real_pos, pos, P * 3);
    h->real_pos_pending = false;   // (a closed-loop position handed over before this call is superseded)
    h->upload(D.start_pos, pos, P * 3);
    // CfAgent::setPosition = clear + push_back for every predicted agent
    h->paths_gen++;
    pmaf_k_launch_restart_paths(D, D.start_pos, h->stream);
    HIP_CHECK(hipGetLastError());
    h->real_pos_h.assign(pos, pos + P * 3);
    for (int p = 0; p < P; p++)  // RealCfAgent::setPosition = push_back
      h->real_path[p].insert(h->real_path[p].end(), {pos[p * 3], pos[p * 3 + 1], pos[p * 3 + 2]});
    sync(h);
    h->scores_valid = false;
    h->rollout_pending = true;
    h->stepped = false;
  });
}

int pmaf_set_real_position(pmaf_planner *h, const double *pos) {
  return guarded([&] {
    REQUIRE(h && pos, "pmaf_set_real_position: NULL argument");
    check_range(pos, (size_t)h->D.P * 3, "pmaf_set_real_position");
    if (h->tick_abandoned)   // (its manager kernel may not have read the staging buffer yet)
      fail(PMAF_ERR_STATE, "pmaf_set_real_position: the previous tick ran into its time limit; call pmaf_stop() first");
    // RealCfAgent::setPosition = push_back (B/src/cf_agent.cpp:44-46): the measured position becomes the real agent's
    // latest one. Handed to the next manager launch through mapped pinned memory -- no wait for the running rollout,
    // no copy command (the node calls this in front of every tick when open_loop is false,
    // B/src/panda_bimanual_control.cpp:333-335)
    std::memcpy(h->h_rp, pos, sizeof(double) * (size_t)h->D.P * 3);
    h->real_pos_pending = true;
    h->real_pos_h.assign(pos, pos + h->D.P * 3);
    for (int p = 0; p < h->D.P; p++)
      h->real_path[p].insert(h->real_path[p].end(), {pos[p * 3], pos[p * 3 + 1], pos[p * 3 + 2]});
  });
}

int pmaf_start(pmaf_planner *h) {
  return guarded([&] {
    REQUIRE(h, "pmaf_start: NULL handle");
    h->use_device();
    if (h->stepped)
      fail(PMAF_ERR_STATE, "pmaf_start: the agents were moved / set by the stepping API (pmaf_move_agent(s), pmaf_set_agent_*); "
                           "call pmaf_reset_agents or pmaf_set_initial_position first (rollouts start from the population's reset state)");
    // a finished rollout that was not reset has nothing left to predict
    // (guard B/src/cf_agent.cpp:310-311 is already false)
    if (!h->rollout_pending) return;
    if (!h->cp_valid) {  // no evaluate yet: score with neutral workspace terms, rescored on evaluate
      h->cp = CostParams{};
      h->cp.ws[0] = h->cp.ws[2] = h->cp.ws[4] = INFINITY;
      h->cp.ws[1] = h->cp.ws[3] = h->cp.ws[5] = -INFINITY;
    }
    bool had_cp = h->cp_valid;
    launch_rollout(h);
    h->cp_valid = had_cp;
    if (!had_cp) h->scores_valid = false;
  });
}

int pmaf_stop(pmaf_planner *h) {
  return guarded([&] {
    REQUIRE(h, "pmaf_stop: NULL handle");
    h->use_device();
    sync(h);
    h->tick_abandoned = false;   // the stream is empty: an abandoned tick's kernels are through
  });
}

// a call that ran into the time limit left its kernels queued: they still read the staging buffers and write the mailbox
static void require_drained(pmaf_planner *h, const char *who) {
  if (h->tick_abandoned)
    fail(PMAF_ERR_STATE, std::string(who) + ": a previous call ran into its time limit with its kernels still queued; drain the "
                         "stream with pmaf_stop() (or recreate the handle) first");
}

int pmaf_evaluate(pmaf_planner *h, const double *cost_gains, const double *ws, int32_t *best_idx) {
  return guarded([&] {
    REQUIRE(h && cost_gains && ws, "pmaf_evaluate: NULL argument");
    h->use_device();
    require_drained(h, "pmaf_evaluate");
    set_cost_params(h, cost_gains, ws);
    ensure_scores(h);
    ManagerArgs A{};
    A.do_select = 1;
    A.out = h->d_out;
    if (h->x.c) {
      A.winner_hdr = claim_exchange_slot(h).d_send;  // (its exchange of two selections ago is through)
      A.winner_stride = (int)winner_rec(h);
    }
    // (sequence numbers of the manager launches that are not ticks count down from -1; the host waits for the mailbox,
    // as pmaf_tick does, instead of for the stream: the result is on the host when the selection is published)
    A.seq = (h->call_seq -= 1.0);
    if (h->h_wp) {   // the winner path of this selection
      A.wp_out = h->d_wp; A.wp_hdr = h->d_wph;
      h->wp_seq = A.seq;
      h->wp_timed = true;
    }
    launch_manager(h, A);
    wait_mailbox(h, A.seq, "pmaf_evaluate");
    h->has_best_h = 1;
    if (h->x.c) begin_exchange(h, h->D.paths);
    if (!h->ev_inflight.empty()) drain_events(h, false);
    refresh_real_cache(h);
    if (best_idx)
      for (int p = 0; p < h->D.P; p++) best_idx[p] = (int32_t)h->h_out[p * PMAF_MBOX];
  });
}

int pmaf_move_real(pmaf_planner *h, const double *obstacles, double dt, int32_t steps, const int32_t *agent_id) {
  return guarded([&] {
    REQUIRE(h && agent_id, "pmaf_move_real: NULL argument");
    REQUIRE(steps >= 0, "pmaf_move_real: steps must be >= 0");
    h->use_device();
    require_drained(h, "pmaf_move_real");
    if (h->has_best_h < 0) {   // unknown (restored / set from outside): read it back once
      sync(h);
      int32_t hb = 0;
      h->download(&hb, h->D.has_best, 1);
      h->has_best_h = hb ? 1 : 0;
    }
    if (!h->has_best_h) fail(PMAF_ERR_STATE, "pmaf_move_real: no best agent yet (call pmaf_evaluate first; the reference dereferences a null best_agent_ here)");
    for (int p = 0; p < h->D.P; p++) REQUIRE(agent_id[p] >= 0 && agent_id[p] < h->D.N, "pmaf_move_real: agent_id out of range");
    // the call's small inputs travel like pmaf_tick's: the obstacle list through the mapped pinned buffer (not at all when
    // it is the resident one), the agent indices by value in the kernel arguments (<= 4 populations), the result through
    // the mailbox -- no copy command and no stream synchronisation (round 5; the node's five calls 106 -> ~45 us)
    // steps == 0: no manager launch reads the list, so it is not staged either (staging marks it resident; a later call
    // with the same list would then hand nothing over and plan on the old obstacles). The reference's moveRealEEAgent
    // passes the list to cfPlanner inside its steps loop only (B/src/cf_manager.cpp:257-263): zero steps leave no trace.
    if (steps == 0) return;
    ResidentGuard resident_guard{h};
    const double *live = stage_live_obstacles_zero_copy(h, obstacles);
    const bool inl = h->D.P <= PMAF_RP_INLINE;
    if (!inl) h->upload(h->d_agent_id, agent_id, h->D.P);
    for (int s = 0; s < steps; s++) {
      ManagerArgs A{};
      A.do_move = 1;
      A.dt_real = dt;
      if (inl) { A.agent_id_inline = 1; for (int p = 0; p < h->D.P; p++) A.agent_id_val[p] = agent_id[p]; }
      else A.agent_id = h->d_agent_id;
      A.live_src = (s == 0) ? live : nullptr;
      A.out = h->d_out;
      A.seq = (h->call_seq -= 1.0);
      launch_manager(h, A);
      wait_mailbox(h, A.seq, "pmaf_move_real");
      resident_guard.armed = false;   // the first launch has read the list (later ones use D.obs_live)
      refresh_real_cache(h);
      append_real_path(h);
    }
  });
}

int pmaf_reset_agents(pmaf_planner *h, const double *pos, const double *vel, const double *obstacles) {
  return guarded([&] {
    REQUIRE(h && pos && vel, "pmaf_reset_agents: NULL argument");
    check_range(pos, (size_t)h->D.P * 3, "pmaf_reset_agents: pos");
    check_range(vel, (size_t)h->D.P * 3, "pmaf_reset_agents: vel");
    h->use_device();
    require_drained(h, "pmaf_reset_agents");
    // an exchange in flight still reads the scored path buffer until its pack kernel is through (the reset rewrites the
    // paths' first points); everything else is ordered by the stream
    for (auto &sl : h->x.slot)
      if (sl.pack_pending) { HIP_CHECK(hipEventSynchronize(sl.ev_pack)); sl.pack_pending = false; }
    ResidentGuard resident_guard{h};
    const double *live = stage_live_obstacles_zero_copy(h, obstacles);
    ManagerArgs A{};
    A.do_reset = 1;
    if (h->D.P <= PMAF_RP_INLINE) {
      A.reset_in_inline = 1;
      for (int p = 0; p < h->D.P; p++)
        for (int c = 0; c < 3; c++) { A.reset_in_val[p * 6 + c] = pos[p * 3 + c]; A.reset_in_val[p * 6 + 3 + c] = vel[p * 3 + c]; }
    } else {
      std::vector<double> in(h->D.P * 6);
      for (int p = 0; p < h->D.P; p++)
        for (int c = 0; c < 3; c++) { in[p * 6 + c] = pos[p * 3 + c]; in[p * 6 + 3 + c] = vel[p * 3 + c]; }
      h->upload(h->d_reset_in, in.data(), in.size());
    }
    A.reset_in = h->d_reset_in;
    A.live_src = live;
    A.out = h->d_out;
    A.seq = (h->call_seq -= 1.0);
    // The mailbox is published once the kernel has read EVERY input of this call (the pinned obstacle list, d_reset_in:
    // both are loaded in front of the publication in k_manager) -- the staging buffers are free again -- and BEFORE its
    // reset stores (obs_start, known_start, the agents' state). Invariant the host relies on for those: whatever reads
    // or overwrites them next is either a launch on h->stream (ordered behind this kernel) or a getter that calls sync().
    launch_manager(h, A);
    wait_mailbox(h, A.seq, "pmaf_reset_agents");
    resident_guard.armed = false;
    refresh_real_cache(h);
    h->scores_valid = false;
    h->rollout_pending = true;
    h->stepped = false;
  });
}

int pmaf_tick(pmaf_planner *h, const double *obstacles, double dt, const double *cost_gains, const double *ws,
              int32_t *best_idx, double *next_pos, double *next_vel) {
  return guarded([&] {
    REQUIRE(h && cost_gains && ws, "pmaf_tick: NULL argument");
    const auto t_entry = std::chrono::steady_clock::now();
    h->use_device();
    require_drained(h, "pmaf_tick");
    // whatever fails between here and the manager kernel's result: the list handed over with this call may not have
    // reached D.obs_live, so it must not count as resident (a retry with the same list has to hand it over again)
    ResidentGuard resident_guard{h};
    set_cost_params(h, cost_gains, ws);
    ensure_scores(h);
    ManagerArgs A{};
    A.live_src = stage_live_obstacles_zero_copy(h, obstacles);
    A.do_select = 1; A.do_move = 1; A.do_reset = 1; A.reset_from_real = 1;
    A.rollout_follows = 1;
    A.dt_real = dt;
    A.out = h->d_out;
    const double want_seq = (double)(++h->mailbox_seq);
    A.seq = h->dbg_withhold ? 0.0 : want_seq;   // (fault injection: pmaf_debug_withhold_mailbox)
    if (h->h_wp) { A.wp_out = h->d_wp; A.wp_hdr = h->d_wph; h->wp_seq = A.seq; h->wp_timed = false; h->last_tick_entry = t_entry; }
    if (h->x.c) {
      // the send buffer of the exchange two ticks back (long through); the previous tick's collective is NOT waited for
      A.winner_hdr = claim_exchange_slot(h).d_send;
      A.winner_stride = (int)winner_rec(h);
    }
    if (h->peer.on) {
      A.peer = h->peer.d_view;
      A.peer_tick = (double)(++h->peer.tick);
    }
    launch_manager(h, A);
    const double *scored = h->D.paths;  // the paths this selection scored; the rollout below writes the other buffer
    h->rollout_pending = true;
    h->stepped = false;
    launch_rollout(h);
    const auto t_enq = std::chrono::steady_clock::now();
    // outputs of k_manager land in mapped pinned memory; wait for them only
    // (no event between the two launches: the host polls the sequence number -- spinning, or with
    // PMAF_FLAG_BLOCKING_WAIT sleeping between polls)
    wait_mailbox(h, want_seq);
    {
      const auto t_sp = std::chrono::steady_clock::now();
      if (h->tick_enq_us.empty()) { h->tick_enq_us.resize(pmaf_planner::TICK_RING); h->tick_sp_us.resize(pmaf_planner::TICK_RING); }
      h->tick_enq_us[h->tick_head] = std::chrono::duration<float, std::micro>(t_enq - t_entry).count();
      h->tick_sp_us[h->tick_head] = std::chrono::duration<float, std::micro>(t_sp - t_entry).count();
      h->tick_head = (h->tick_head + 1) % pmaf_planner::TICK_RING;
      if (h->tick_count < pmaf_planner::TICK_RING) h->tick_count++;
    }
    resident_guard.armed = false;   // the manager kernel has read the list and published its result
    h->has_best_h = 1;
    const int peer_late = h->peer.on ? peer_book_tick(h) : 0;
    if (h->x.c) begin_exchange(h, scored);  // pack + all-gather on the exchange stream, beside the rollout
    refresh_real_cache(h);
    append_real_path(h);
    for (int p = 0; p < h->D.P; p++) {
      const double *o = h->h_out + p * PMAF_MBOX;
      if (best_idx) best_idx[p] = (int32_t)o[0];
      if (next_pos) { next_pos[p * 3] = o[1]; next_pos[p * 3 + 1] = o[2]; next_pos[p * 3 + 2] = o[3]; }
      if (next_vel) { next_vel[p * 3] = o[4]; next_vel[p * 3 + 1] = o[5]; next_vel[p * 3 + 2] = o[6]; }
    }
    if (peer_late == 1)
      fail(PMAF_ERR_DEVICE, "pmaf_tick: the header of a coupled peer population did not arrive in time (a peer rank behind "
                            "by more than PMAF_PEER_TIMEOUT_S, or gone); the trailing obstacle kept its previous value");
    if (peer_late == 2)
      fail(PMAF_ERR_STATE, "pmaf_tick: a coupled peer population had already overwritten the header this tick needs (it ran "
                           "ahead): couplings must be pairwise mutual and every rank must issue the same pmaf_tick calls "
                           "(include/pmaf.h, peer mailboxes: TOPOLOGY); the trailing obstacle kept its previous value");
    if (peer_late == 3)
      fail(PMAF_ERR_STATE, "pmaf_tick: a coupled peer population is not coupled back to this one -- couplings must be "
                           "pairwise mutual (include/pmaf.h, peer mailboxes: TOPOLOGY)");
  });
}

// ---- synchronous stepping API ----
static const double *upload_plan_obstacles(pmaf_planner *h, const double *obstacles) {
  const size_t n = (size_t)h->D.P * 7 * h->D.n_obs;
  check_range(obstacles, (size_t)h->D.P * h->D.n_obs * 7, "obstacles");
  if (!h->d_plan_obs) {
    h->d_plan_obs = h->dalloc_untracked<double>(n);
    h->d_plan_out = h->dalloc_untracked<double>((size_t)h->D.P * h->D.N);
    h->d_plan_calls = h->dalloc_untracked<int32_t>((size_t)h->D.P);
  }
  std::vector<double> soa(n);
  aos_to_soa(obstacles, soa.data(), h->D.P, h->D.n_obs);
  h->upload(h->d_plan_obs, soa.data(), n);
  return h->d_plan_obs;
}

static void plan_steps(pmaf_planner *h, const double *obstacles, double dt, int steps, const int32_t *agent_id,
                       int max_calls, int32_t *calls) {
  check_range(&dt, 1, "dt");
  h->use_device();
  sync(h);
  const DevView &D = h->D;
  std::vector<int32_t> np((size_t)D.P * D.N);
  h->download(np.data(), D.n_points, np.size());
  if (!agent_id) {  // moveAgents: every path must have room for `steps` more points
    int mx = 0;
    for (int32_t v : np) mx = v > mx ? v : mx;
    if ((long)mx + steps > (long)D.cap)
      fail(PMAF_ERR_STATE, "pmaf_move_agents: a path would outgrow max_prediction_steps (the path buffers are fixed-size)");
  }
  PlanArgs A{};
  A.obs = upload_plan_obstacles(h, obstacles);
  A.dt = dt;
  A.steps = steps;
  A.max_calls = agent_id ? max_calls : 1;
  A.until_goal = agent_id ? 1 : 0;
  if (agent_id) {
    for (int p = 0; p < D.P; p++) REQUIRE(agent_id[p] >= 0 && agent_id[p] < D.N, "pmaf_move_agent: agent_id out of range");
    h->upload(h->d_agent_id, agent_id, D.P);
    A.only = h->d_agent_id;
    A.calls_out = h->d_plan_calls;
    HIP_CHECK(hipMemsetAsync(h->d_plan_calls, 0, sizeof(int32_t) * D.P, h->stream));
  }
  h->paths_gen++;
  if (!pmaf_k_launch_plan_steps(D, A, h->lpa, h->n_blocks, h->lds_rollout, h->stream)) fail(PMAF_ERR_INVALID, "bad lanes_per_agent");
  HIP_CHECK(hipGetLastError());
  sync(h);
  if (agent_id && calls) h->download(calls, h->d_plan_calls, D.P);
  h->scores_valid = false;     // path costs are re-scored on demand (k_score)
  h->rollout_pending = false;
  h->stepped = true;
}

int pmaf_move_agents(pmaf_planner *h, const double *obstacles, double dt, int32_t steps) {
  return guarded([&] {
    REQUIRE(h && obstacles, "pmaf_move_agents: NULL argument");
    REQUIRE(steps >= 0, "pmaf_move_agents: steps must be >= 0");
    plan_steps(h, obstacles, dt, steps, nullptr, 1, nullptr);
  });
}

int pmaf_move_agent(pmaf_planner *h, const double *obstacles, double dt, int32_t steps, const int32_t *agent_id,
                    int32_t max_calls, int32_t *calls) {
  return guarded([&] {
    REQUIRE(h && obstacles && agent_id, "pmaf_move_agent: NULL argument");
    REQUIRE(steps >= 1 && max_calls >= 0, "pmaf_move_agent: need steps >= 1 and max_calls >= 0");
    plan_steps(h, obstacles, dt, steps, agent_id, max_calls, calls);
  });
}

static void set_agents(pmaf_planner *h, const double *pos, const double *vel) {
  const int P = h->D.P;
  check_range(pos, (size_t)P * 3, "pos");
  if (vel) check_range(vel, (size_t)P * 3, "vel");
  h->use_device();
  sync(h);
  std::vector<double> in((size_t)P * 6, 0.0);
  for (int p = 0; p < P; p++)
    for (int c = 0; c < 3; c++) { in[p * 6 + c] = pos[p * 3 + c]; if (vel) in[p * 6 + 3 + c] = vel[p * 3 + c]; }
  // d_reset_in holds [P][6]; the kernel takes two [P][3] arrays: repack
  std::vector<double> pk((size_t)P * 6);
  for (int p = 0; p < P; p++)
    for (int c = 0; c < 3; c++) { pk[p * 3 + c] = in[p * 6 + c]; pk[(size_t)P * 3 + p * 3 + c] = in[p * 6 + 3 + c]; }
  h->upload(h->d_reset_in, pk.data(), pk.size());
  h->paths_gen++;
  pmaf_k_launch_set_agents(h->D, h->d_reset_in, vel ? h->d_reset_in + (size_t)P * 3 : nullptr, h->stream);
  HIP_CHECK(hipGetLastError());
  sync(h);
  h->scores_valid = false;
  h->rollout_pending = false;
  // like after pmaf_move_agent(s): in the reference a startPrediction() here would continue with each agent's OWN
  // velocity, known flags and advanced obstacle copies (CfAgent::setPosition only clears the path,
  // B/src/cf_agent.cpp:39-42), which a rollout launched from the population's reset state cannot reproduce
  h->stepped = true;
}

int pmaf_set_agent_positions(pmaf_planner *h, const double *pos) {
  return guarded([&] {
    REQUIRE(h && pos, "pmaf_set_agent_positions: NULL argument");
    set_agents(h, pos, nullptr);
  });
}

int pmaf_set_agent_pos_and_vels(pmaf_planner *h, const double *pos, const double *vel) {
  return guarded([&] {
    REQUIRE(h && pos && vel, "pmaf_set_agent_pos_and_vels: NULL argument");
    set_agents(h, pos, vel);
  });
}

int pmaf_eval_obstacle_distance(pmaf_planner *h, const double *obstacles, double *out) {
  return guarded([&] {
    REQUIRE(h && obstacles && out, "pmaf_eval_obstacle_distance: NULL argument");
    h->use_device();
    sync(h);
    const double *d_obs = upload_plan_obstacles(h, obstacles);
    pmaf_k_launch_eval_obstacle_distance(h->D, d_obs, h->d_plan_out, h->stream);
    HIP_CHECK(hipGetLastError());
    h->download(out, h->d_plan_out, (size_t)h->D.P * h->D.N);
  });
}

int pmaf_link_force(pmaf_planner *h, int32_t pop, int32_t n, const double *link_pos, const double *k_r_force,
                    const double *obstacles, double *out) {
  return guarded([&] {
    REQUIRE(h && link_pos && k_r_force && obstacles && out, "pmaf_link_force: NULL argument");
    REQUIRE(pop >= 0 && pop < h->D.P && n >= 0, "pmaf_link_force: bad population or count");
    if (n == 0) return;
    const double *sent = obstacles + ((size_t)pop * h->D.n_obs + (h->D.n_obs - 1)) * 7;
    check_range(link_pos, (size_t)n * 3, "pmaf_link_force: link_pos");
    check_range(k_r_force, (size_t)n, "pmaf_link_force: k_r_force");
    check_range(sent, 7, "pmaf_link_force: obstacles (last)");
    h->use_device();
    // per-handle scratch, grown on demand: [3n] link points | [n] gains | [7] obstacle | [3n] forces
    const size_t need = (size_t)n * 7 + 7;
    if (need > h->link_scratch_doubles) {
      sync(h);
      if (h->d_link) { (void)hipFree(h->d_link); h->d_link = nullptr; h->link_scratch_doubles = 0; }
      if (h->h_link) { (void)hipHostFree(h->h_link); h->h_link = nullptr; }
      const size_t cap_d = need < 512 ? 512 : need * 2;
      HIP_CHECK(hipMalloc((void **)&h->d_link, sizeof(double) * cap_d));
      HIP_CHECK(hipHostMalloc((void **)&h->h_link, sizeof(double) * cap_d, hipHostMallocDefault));
      h->link_scratch_doubles = cap_d;
    }
    double *hb = h->h_link, *db = h->d_link;
    std::memcpy(hb, link_pos, sizeof(double) * 3 * n);
    std::memcpy(hb + 3 * (size_t)n, k_r_force, sizeof(double) * n);
    std::memcpy(hb + 4 * (size_t)n, sent, sizeof(double) * 7);
    const size_t in_d = 4 * (size_t)n + 7;
    HIP_CHECK(hipMemcpyAsync(db, hb, sizeof(double) * in_d, hipMemcpyHostToDevice, h->stream));
    pmaf_k_launch_link_force(n, db, db + 3 * (size_t)n, db + 4 * (size_t)n, h->D.C.rad, h->D.C.shell, db + in_d, h->stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(hb + in_d, db + in_d, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, h->stream));
    sync(h);
    std::memcpy(out, hb + in_d, sizeof(double) * 3 * n);
  });
}

#define GETTER_PROLOGUE(name)                    \
  REQUIRE(h, name ": NULL handle");              \
  h->use_device();                               \
  sync(h);                                       \
  const DevView &D = h->D;                       \
  const size_t PN = (size_t)D.P * D.N;           \
  (void)PN;

// bring the host mirror of paths / n_points up to date (one D2H per rollout generation)
static void refresh_paths_mirror(pmaf_planner *h) {
  if (h->mirror_gen == h->paths_gen) return;
  const DevView &D = h->D;
  const size_t PN = (size_t)D.P * D.N;
  if (!h->h_paths) {
    HIP_CHECK(hipHostMalloc((void **)&h->h_paths, sizeof(double) * PN * (size_t)D.cap * 3, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **)&h->h_np, sizeof(int32_t) * PN, hipHostMallocDefault));
  }
  HIP_CHECK(hipMemcpyAsync(h->h_np, D.n_points, sizeof(int32_t) * PN, hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(hipMemcpyAsync(h->h_paths, D.paths, sizeof(double) * PN * (size_t)D.cap * 3, hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  // entries past an agent's path end are stale device memory: report zeros
  for (size_t pa = 0; pa < PN; pa++)
    std::memset(h->h_paths + (pa * D.cap + h->h_np[pa]) * 3, 0, sizeof(double) * 3 * (size_t)(D.cap - h->h_np[pa]));
  h->mirror_gen = h->paths_gen;
}

int pmaf_get_paths(pmaf_planner *h, double *paths, int32_t *n_points) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_paths")
    refresh_paths_mirror(h);
    if (paths) std::memcpy(paths, h->h_paths, sizeof(double) * PN * (size_t)D.cap * 3);
    if (n_points) std::memcpy(n_points, h->h_np, sizeof(int32_t) * PN);
  });
}
// zero-copy view of the same mirror: *paths [P][N][cap][3] and *n_points [P][N] stay valid until the next call that
// changes the predicted paths (start / tick / reset / set_* / move_agents / load_state)
int pmaf_view_paths(pmaf_planner *h, const double **paths, const int32_t **n_points) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_view_paths")
    refresh_paths_mirror(h);
    if (paths) *paths = h->h_paths;
    if (n_points) *n_points = h->h_np;
  });
}
int pmaf_get_costs(pmaf_planner *h, double *costs) {
  return guarded([&] { GETTER_PROLOGUE("pmaf_get_costs") REQUIRE(costs, "NULL out"); h->download(costs, D.costs, PN); });
}
int pmaf_get_path_lengths(pmaf_planner *h, double *out) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_path_lengths")
    REQUIRE(out, "NULL out");
    if (!h->cp_valid) {
      h->cp = CostParams{};
      h->cp.ws[0] = h->cp.ws[2] = h->cp.ws[4] = INFINITY;
      h->cp.ws[1] = h->cp.ws[3] = h->cp.ws[5] = -INFINITY;
    }
    bool had = h->cp_valid;
    ensure_scores(h);
    if (!had) h->scores_valid = false;
    sync(h);
    h->download(out, D.path_len, PN);
  });
}
int pmaf_get_min_obs_dist(pmaf_planner *h, double *out) {
  return guarded([&] { GETTER_PROLOGUE("pmaf_get_min_obs_dist") REQUIRE(out, "NULL out"); h->download(out, D.min_obs, PN); });
}
int pmaf_get_success(pmaf_planner *h, int32_t *out) {
  return guarded([&] { GETTER_PROLOGUE("pmaf_get_success") REQUIRE(out, "NULL out"); h->download(out, D.reached, PN); });
}
int pmaf_get_agent_velocities(pmaf_planner *h, double *out) {
  return guarded([&] { GETTER_PROLOGUE("pmaf_get_agent_velocities") REQUIRE(out, "NULL out"); h->download(out, D.agent_vel, PN * 3); });
}
int pmaf_get_rotation_vectors(pmaf_planner *h, double *rot, int32_t *known) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_rotation_vectors")
    const int n_obs = D.n_obs;
    if (rot) {
      std::vector<double> r(PN * 3 * n_obs);
      h->download(r.data(), D.rot, r.size());
      for (size_t pa = 0; pa < PN; pa++)
        for (int i = 0; i < n_obs; i++)
          for (int c = 0; c < 3; c++) rot[(pa * n_obs + i) * 3 + c] = r[(pa * 3 + c) * n_obs + i];
    }
    if (known) h->download(known, D.known_out, PN * n_obs);
  });
}
int pmaf_get_real_state(pmaf_planner *h, double *pos, double *vel, double *force) {
  return guarded([&] {
    // served from the host copy kept current by every call that changes the real
    // agent: no wait for the running rollout (the reference's getters are instant)
    REQUIRE(h, "pmaf_get_real_state: NULL handle");
    const size_t n = sizeof(double) * 3 * (size_t)h->D.P;
    if (pos) std::memcpy(pos, h->real_pos_h.data(), n);
    if (vel) std::memcpy(vel, h->real_vel_h.data(), n);
    if (force) std::memcpy(force, h->real_force_h.data(), n);
  });
}
int pmaf_get_real_known(pmaf_planner *h, int32_t *known, double *rot) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_real_known")
    const int n_obs = D.n_obs;
    if (known) h->download(known, D.real_known, (size_t)D.P * n_obs);
    if (rot) {
      std::vector<double> r((size_t)D.P * 3 * n_obs);
      h->download(r.data(), D.real_rot, r.size());
      for (int p = 0; p < D.P; p++)
        for (int i = 0; i < n_obs; i++)
          for (int c = 0; c < 3; c++) rot[((size_t)p * n_obs + i) * 3 + c] = r[((size_t)p * 3 + c) * n_obs + i];
    }
  });
}
int pmaf_get_real_path(pmaf_planner *h, int32_t pop, double *out, int32_t max_points, int32_t *n_total) {
  return guarded([&] {
    REQUIRE(h && pop >= 0 && pop < h->D.P, "pmaf_get_real_path: bad argument");
    const std::vector<double> &rp = h->real_path[pop];
    int n = (int)(rp.size() / 3);
    if (n_total) *n_total = n;
    if (out && max_points > 0) std::memcpy(out, rp.data(), sizeof(double) * 3 * (size_t)(n < max_points ? n : max_points));
  });
}
int pmaf_get_dist_from_goal(pmaf_planner *h, double *out) {
  return guarded([&] {
    REQUIRE(h && out, "pmaf_get_dist_from_goal: NULL argument");
    // (goal_pos_ - real.getLatestPosition()).norm(), cf_manager.h:87-89, from the
    // host copy of the real position; same operation order as the device code
    const std::vector<double> &rp = h->real_pos_h;
    for (int p = 0; p < h->D.P; p++) {
      double dx = h->goal_h[p * 3] - rp[p * 3], dy = h->goal_h[p * 3 + 1] - rp[p * 3 + 1], dz = h->goal_h[p * 3 + 2] - rp[p * 3 + 2];
#ifdef PMAF_DOT_RIGHT_ASSOC
      out[p] = std::sqrt(dx * dx + (dy * dy + dz * dz));
#else
      out[p] = std::sqrt((dx * dx + dy * dy) + dz * dz);
#endif
    }
  });
}
int pmaf_get_best(pmaf_planner *h, int32_t *type, int32_t *id) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_best")
    std::vector<int32_t> hb(D.P), bt(D.P), bi(D.P);
    h->download(hb.data(), D.has_best, D.P);
    h->download(bt.data(), D.best_type, D.P);
    h->download(bi.data(), D.best_id, D.P);
    for (int p = 0; p < D.P; p++) {
      if (type) type[p] = hb[p] ? bt[p] : -1;
      if (id) id[p] = hb[p] ? bi[p] : 0;
    }
  });
}
int pmaf_get_prediction_times_ns(pmaf_planner *h, double *out) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_prediction_times_ns")
    REQUIRE(out, "NULL out");
    // per-agent device clock (wall_clock64: constant-rate counter, rate in kHz)
    int khz = 0;
    HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device));
    if (khz <= 0) khz = 100000;
    std::vector<unsigned long long> t(PN);
    h->download(t.data(), D.pred_ticks, PN);
    for (size_t i = 0; i < PN; i++) out[i] = (double)t[i] * (1e6 / (double)khz);
  });
}

int pmaf_set_best(pmaf_planner *h, const int32_t *id, const int32_t *type, const double *rand_vecs) {
  return guarded([&] {
    REQUIRE(h && id && type, "pmaf_set_best: NULL argument");
    h->use_device();
    sync(h);
    const DevView &D = h->D;
    std::vector<int32_t> hb(D.P);
    for (int p = 0; p < D.P; p++) {
      REQUIRE(id[p] >= 0 && id[p] <= D.N, "pmaf_set_best: id out of range");
      REQUIRE(id[p] == 0 || (type[p] >= PMAF_GOAL_HEURISTIC && type[p] <= PMAF_HAD_HEURISTIC),
              "pmaf_set_best: type must be one of the six heuristics when id > 0");
      hb[p] = id[p] > 0;
      if (p == 0) h->has_best_h = hb[p] ? 1 : 0;
    }
    h->upload(D.has_best, hb.data(), D.P);
    h->upload(D.best_id, id, D.P);
    h->upload(D.best_type, type, D.P);
    if (rand_vecs) {
      const int n_obs = D.n_obs;
      check_range(rand_vecs, (size_t)D.P * n_obs * 3, "pmaf_set_best: rand_vecs");
      std::vector<double> r((size_t)D.P * 3 * n_obs);
      for (int p = 0; p < D.P; p++)
        for (int i = 0; i < n_obs; i++)
          for (int c = 0; c < 3; c++) r[((size_t)p * 3 + c) * n_obs + i] = rand_vecs[((size_t)p * n_obs + i) * 3 + c];
      h->upload(D.best_rnd, r.data(), r.size());
    }
  });
}

size_t pmaf_winner_record_doubles(const pmaf_planner *h) { return h ? winner_rec(h) : 0; }

int pmaf_write_winner_records(pmaf_planner *h, void *dst_device, size_t bytes) {
  return guarded([&] {
    REQUIRE(h && dst_device, "pmaf_write_winner_records: NULL argument");
    REQUIRE(bytes >= sizeof(double) * pmaf_winner_record_doubles(h) * h->D.P, "pmaf_write_winner_records: buffer too small");
    h->use_device();
    flush_real_position(h);
    pmaf_k_launch_winner(h->D, (double *)dst_device, h->stream);
    HIP_CHECK(hipGetLastError());
  });
}

int pmaf_allgather_winners(pmaf_planner *h, pmaf_comm *c, void *recv_device, size_t bytes) {
  return guarded([&] {
    REQUIRE(h && c && recv_device, "pmaf_allgather_winners: NULL argument");
    const size_t n_local = (size_t)h->D.P * winner_rec(h);
    REQUIRE(bytes >= sizeof(double) * n_local * (size_t)c->world, "pmaf_allgather_winners: receive buffer too small");
    h->use_device();
    if (!h->d_send1) HIP_CHECK(hipMalloc((void **)&h->d_send1, sizeof(double) * n_local));
    flush_real_position(h);
    pmaf_k_launch_winner(h->D, h->d_send1, h->stream);
    HIP_CHECK(hipGetLastError());
    if (c->rccl) {
      // stream-ordered behind the pack kernel: no host synchronisation between the two
      const std::string err = pmaf_comm_enqueue_allgather(c, h->d_send1, (double *)recv_device, n_local, h->stream);
      if (!err.empty()) fail(PMAF_ERR_DEVICE, err);
    } else {
      std::vector<double> snd(n_local), rcv(n_local * (size_t)c->world);
      h->download(snd.data(), h->d_send1, n_local);
      if (c->fn(c->ctx, snd.data(), rcv.data(), n_local * sizeof(double)) != 0)
        fail(PMAF_ERR_DEVICE, "pmaf_allgather_winners: the host all-gather callback failed");
      h->upload((double *)recv_device, rcv.data(), rcv.size());
    }
  });
}

int pmaf_attach_comm(pmaf_planner *h, pmaf_comm *c) {
  return guarded([&] {
    REQUIRE(h, "pmaf_attach_comm: NULL handle");
    h->use_device();
    sync(h);
    detach_comm(h);
    if (!c) return;
    REQUIRE(!c->rccl || c->device == h->device, "pmaf_attach_comm: the communicator lives on another device than the handle");
    pmaf_planner::Exchange &x = h->x;
    const size_t n_local = (size_t)h->D.P * winner_rec(h);
    const size_t path_bytes = sizeof(double) * (size_t)h->D.P * h->D.N * h->D.cap * 3;
    try {
      {
        // high-priority streams: their own hardware queues, so the pack kernel and the all-gather are dispatched beside
        // the rollout instead of queueing with it (streams of equal priority may share a hardware queue: measured
        // +48 us per C2 tick with the exchange on a default-priority stream)
        int lo = 0, hi = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_CHECK(hipStreamCreateWithPriority(&x.xp, hipStreamNonBlocking, hi));
        HIP_CHECK(hipStreamCreateWithPriority(&x.xs, hipStreamNonBlocking, hi));
      }
      for (auto &sl : x.slot) {
        HIP_CHECK(hipEventCreateWithFlags(&sl.ev_pack, hipEventDisableTiming));
        HIP_CHECK(hipEventCreate(&sl.ev_t0));
        HIP_CHECK(hipEventCreate(&sl.ev_t1));
        HIP_CHECK(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        HIP_CHECK(hipMalloc((void **)&sl.d_send, sizeof(double) * n_local));
        HIP_CHECK(hipMalloc((void **)&sl.d_recv, sizeof(double) * n_local * (size_t)c->world));
        HIP_CHECK(hipHostMalloc((void **)&sl.h_send, sizeof(double) * n_local, hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void **)&sl.h_recv, sizeof(double) * n_local * (size_t)c->world, hipHostMallocDefault));
        HIP_CHECK(hipMemset(sl.d_send, 0, sizeof(double) * n_local));
        HIP_CHECK(hipMemset(sl.d_recv, 0, sizeof(double) * n_local * (size_t)c->world));
        std::memset(sl.h_recv, 0, sizeof(double) * n_local * (size_t)c->world);
      }
      HIP_CHECK(hipMalloc((void **)&x.paths_b, path_bytes));
      x.paths_a = h->D.paths;
      x.c = c;
      h->paths_gen++;
    } catch (...) {
      x.c = c;  // so that detach_comm releases what was created
      x.paths_a = h->D.paths;
      detach_comm(h);
      throw;
    }
  });
}

int pmaf_winners_wait(pmaf_planner *h, const double **records, size_t *n_doubles) {
  return guarded([&] {
    REQUIRE(h, "pmaf_winners_wait: NULL handle");
    if (!h->x.c) fail(PMAF_ERR_STATE, "pmaf_winners_wait: no communicator attached (pmaf_attach_comm)");
    h->use_device();
    finish_exchanges(h);
    if (h->x.failed) fail(PMAF_ERR_DEVICE, "pmaf_winners_wait: the last winner exchange failed (no table)");
    if (records) *records = h->x.slot[h->x.cur].h_recv;
    if (n_doubles) *n_doubles = (size_t)h->D.P * winner_rec(h) * (size_t)h->x.c->world;
  });
}

void *pmaf_winners_device(pmaf_planner *h) { return h ? (void *)h->x.slot[h->x.cur].d_recv : nullptr; }

int pmaf_get_exchange_times_us(pmaf_planner *h, double *out, int32_t max_n, int32_t *n) {
  return guarded([&] {
    REQUIRE(h && n, "pmaf_get_exchange_times_us: NULL argument");
    std::vector<double> &v = h->x.ag_us;
    const size_t k = (out && max_n > 0) ? std::min(v.size(), (size_t)max_n) : 0;
    for (size_t i = 0; i < k; i++) out[i] = v[i];
    *n = (int32_t)k;
    v.clear();
  });
}

int pmaf_get_tick_times_us(pmaf_planner *h, double *enqueue_us, double *setpoint_us, int32_t max_n, int32_t *n) {
  return guarded([&] {
    REQUIRE(h && n, "pmaf_get_tick_times_us: NULL argument");
    const size_t have = h->tick_count;
    const size_t k = (max_n > 0) ? std::min(have, (size_t)max_n) : 0;
    // oldest first among the newest k
    for (size_t i = 0; i < k; i++) {
      const size_t at = (h->tick_head + pmaf_planner::TICK_RING - k + i) % pmaf_planner::TICK_RING;
      if (enqueue_us) enqueue_us[i] = h->tick_enq_us[at];
      if (setpoint_us) setpoint_us[i] = h->tick_sp_us[at];
    }
    *n = (int32_t)k;
    h->tick_count = 0;
    h->tick_head = 0;
  });
}

// ---- peer mailboxes (include/pmaf.h) ----
int pmaf_peer_export(pmaf_planner *h, int32_t world, void *handle_out) {
  return guarded([&] {
    REQUIRE(h && handle_out, "pmaf_peer_export: NULL argument");
    REQUIRE(world >= 1 && world <= PMAF_PEER_MAX_WORLD, "pmaf_peer_export: need 1 <= world <= 64");
    h->use_device();
    sync(h);
    peer_disconnect(h);
    pmaf_planner::Peer &pr = h->peer;
    const size_t bytes = sizeof(double) * peer_inbox_doubles(world, h->D.P);
    // fine-grained device memory: stores of other agents (peer GPUs over xGMI) become visible to this GPU's
    // system-scope loads without a kernel boundary; plain device memory as the fall-back
    hipError_t e = hipExtMallocWithFlags((void **)&pr.inbox, bytes, hipDeviceMallocFinegrained);
    pr.inbox_fine = e == hipSuccess;
    if (e != hipSuccess) { (void)hipGetLastError(); HIP_CHECK(hipMalloc((void **)&pr.inbox, bytes)); }
    {
      // sequence numbers start at -1 (no header yet)
      std::vector<double> init(peer_inbox_doubles(world, h->D.P), 0.0);
      for (size_t i = 8; i < init.size(); i += PMAF_PEER_SLOT) init[i] = -1.0;
      HIP_CHECK(hipMemcpy(pr.inbox, init.data(), bytes, hipMemcpyHostToDevice));
    }
    pr.world = world;
    PeerHandleBlob b{};
    b.magic = kPeerMagic; b.pid = (int64_t)getpid(); b.ptr = (uint64_t)(uintptr_t)pr.inbox;
    b.world = world; b.P = h->D.P; b.device = h->device;
    pci_id_of(h->device, b.pci);
    if (world > 1) {
      e = hipIpcGetMemHandle(&b.ipc, pr.inbox);
      if (e != hipSuccess && pr.inbox_fine) {   // this runtime does not export fine-grained memory: plain device memory
        (void)hipGetLastError();
        (void)hipFree(pr.inbox); pr.inbox = nullptr; pr.inbox_fine = false;
        HIP_CHECK(hipMalloc((void **)&pr.inbox, bytes));
        std::vector<double> init(peer_inbox_doubles(world, h->D.P), 0.0);
        for (size_t i = 8; i < init.size(); i += PMAF_PEER_SLOT) init[i] = -1.0;
        HIP_CHECK(hipMemcpy(pr.inbox, init.data(), bytes, hipMemcpyHostToDevice));
        b.ptr = (uint64_t)(uintptr_t)pr.inbox;
        e = hipIpcGetMemHandle(&b.ipc, pr.inbox);
      }
      if (e != hipSuccess) throw HipError{e, "hipIpcGetMemHandle (peer inbox)", __LINE__};
    }
    b.fine = pr.inbox_fine ? 1 : 0;   // (pmaf_peer_connect refuses cross-device peers of a plain-memory inbox)
    std::memcpy(handle_out, &b, sizeof(b));
  });
}

int pmaf_peer_connect(pmaf_planner *h, int32_t world, int32_t rank, const void *handles) {
  return guarded([&] {
    REQUIRE(h && handles, "pmaf_peer_connect: NULL argument");
    pmaf_planner::Peer &pr = h->peer;
    REQUIRE(pr.inbox && !pr.on, "pmaf_peer_connect: call pmaf_peer_export first (once per connection)");
    REQUIRE(world == pr.world && rank >= 0 && rank < world, "pmaf_peer_connect: world differs from pmaf_peer_export's, or bad rank");
    h->use_device();
    sync(h);
    const PeerHandleBlob *hb = static_cast<const PeerHandleBlob *>(handles);
    pr.mapped.assign(world, nullptr);
    pr.opened.assign(world, 0);
    for (int r = 0; r < world; r++) {
      const PeerHandleBlob &b = hb[r];
      REQUIRE(b.magic == kPeerMagic, "pmaf_peer_connect: not a handle of pmaf_peer_export");
      REQUIRE(b.world == world && b.P == h->D.P, "pmaf_peer_connect: every rank must export for the same world and hold the same number of populations");
      if (r != rank) {
        // a peer on ANOTHER GPU stores into / is stored into over xGMI while the kernels run: both inboxes must be
        // fine-grained memory (plain device memory does not make another agent's stores visible to a running kernel --
        // the in-kernel poll would time out every tick with no hint of the cause)
        int32_t mine[3];
        pci_id_of(h->device, mine);
        const bool same_gpu = mine[0] == b.pci[0] && mine[1] == b.pci[1] && mine[2] == b.pci[2] && mine[1] >= 0;
        if (!same_gpu && !(pr.inbox_fine && b.fine))
          fail(PMAF_ERR_DEVICE, "pmaf_peer_connect: a peer on another GPU needs fine-grained inboxes on both sides, and this "
                                "runtime could only export plain device memory (pmaf_peer_info); use the winner-record "
                                "exchange (pmaf_attach_comm) for the coupling instead");
      }
      if (r == rank) {
        REQUIRE(b.pid == (int64_t)getpid() && b.ptr == (uint64_t)(uintptr_t)pr.inbox, "pmaf_peer_connect: handles[rank] is not this handle's own export");
        pr.mapped[r] = pr.inbox;
      } else if (b.pid == (int64_t)getpid()) {
        // another handle of this process (several "ranks" in one process: tests, one host driving several GPUs)
        if (b.device != h->device) {
          hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw HipError{e, "hipDeviceEnablePeerAccess (peer inbox)", __LINE__};
          (void)hipGetLastError();
        }
        pr.mapped[r] = reinterpret_cast<double *>((uintptr_t)b.ptr);
      } else {
        void *p = nullptr;
        HIP_CHECK(hipIpcOpenMemHandle(&p, b.ipc, hipIpcMemLazyEnablePeerAccess));
        pr.mapped[r] = static_cast<double *>(p);
        pr.opened[r] = 1;
      }
    }
    pr.rank = rank;
    const int P = h->D.P;
    HIP_CHECK(hipMalloc((void **)&pr.d_view, sizeof(PeerView)));
    HIP_CHECK(hipMalloc((void **)&pr.d_couple, sizeof(int32_t) * 2 * P));
    HIP_CHECK(hipMalloc((void **)&pr.d_radius, sizeof(double) * P));
    pr.couple_h.assign(2 * (size_t)P, -1);
    pr.radius_h.assign(P, 0.0);
    HIP_CHECK(hipMemcpy(pr.d_couple, pr.couple_h.data(), sizeof(int32_t) * 2 * P, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(pr.d_radius, pr.radius_h.data(), sizeof(double) * P, hipMemcpyHostToDevice));
    int khz = 0;
    HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device));
    if (khz <= 0) khz = 100000;
    pr.us_per_tick = 1e3 / (double)khz;
    double timeout_s = 2.0;
    { const char *to = getenv("PMAF_PEER_TIMEOUT_S"); if (to && atof(to) > 0.0) timeout_s = atof(to); }
    // the in-kernel wait for a peer's header must end before the host's wait for the tick does (PMAF_TICK_TIMEOUT_S):
    // otherwise a late peer is reported as a hung device and the tick is abandoned with its kernel still waiting
    if (timeout_s > 0.5 * h->tick_timeout_s) timeout_s = 0.5 * h->tick_timeout_s;
    PeerView v{};
    v.world = world; v.rank = rank; v.P = P;
    v.inbox = pr.inbox;
    for (int r = 0; r < world; r++) v.peer[r] = pr.mapped[r];
    v.couple = pr.d_couple; v.couple_radius = pr.d_radius;
    v.timeout_ticks = (unsigned long long)(timeout_s * 1e3 * (double)khz);
    HIP_CHECK(hipMemcpy(pr.d_view, &v, sizeof(v), hipMemcpyHostToDevice));
    pr.tick = 0;
    pr.on = true;
    h->live_resident.clear();
  });
}

int pmaf_peer_couple(pmaf_planner *h, int32_t pop, int32_t src_rank, int32_t src_pop, double radius, const double *init_pos) {
  return guarded([&] {
    REQUIRE(h, "pmaf_peer_couple: NULL handle");
    pmaf_planner::Peer &pr = h->peer;
    if (!pr.on) fail(PMAF_ERR_STATE, "pmaf_peer_couple: no peer mailboxes connected (pmaf_peer_connect)");
    REQUIRE(pop >= 0 && pop < h->D.P, "pmaf_peer_couple: bad population");
    h->use_device();
    sync(h);
    if (src_rank >= 0) {
      REQUIRE(src_rank < pr.world && src_pop >= 0 && src_pop < h->D.P, "pmaf_peer_couple: bad source rank / population");
      REQUIRE(!(src_rank == pr.rank && src_pop == pop), "pmaf_peer_couple: a population cannot be coupled to itself");
      check_range(&radius, 1, "pmaf_peer_couple: radius");
      if (init_pos) {
        if (pr.tick != 0) fail(PMAF_ERR_STATE, "pmaf_peer_couple: init_pos can only be given before the first pmaf_tick after pmaf_peer_connect");
        check_range(init_pos, 3, "pmaf_peer_couple: init_pos");
        // header "0" of the source population in this rank's own inbox (parity slot 0; its first real writer is the
        // source's tick 2, which needs this rank's tick 1 first)
        double slot[PMAF_PEER_SLOT] = {0};
        slot[4] = init_pos[0]; slot[5] = init_pos[1]; slot[6] = init_pos[2]; slot[8] = 0.0;
        slot[9] = slot[10] = -2.0;   // host-written header: the publisher's own coupling is not known here (k_manager's mutuality check skips it)
        HIP_CHECK(hipMemcpy(pr.inbox + (((size_t)0 * pr.world + src_rank) * h->D.P + src_pop) * PMAF_PEER_SLOT, slot,
                            sizeof(slot), hipMemcpyHostToDevice));
      }
    }
    pr.couple_h[2 * pop] = src_rank < 0 ? -1 : src_rank;
    pr.couple_h[2 * pop + 1] = src_rank < 0 ? -1 : src_pop;
    pr.radius_h[pop] = src_rank < 0 ? 0.0 : radius;
    HIP_CHECK(hipMemcpy(pr.d_couple, pr.couple_h.data(), sizeof(int32_t) * 2 * h->D.P, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(pr.d_radius, pr.radius_h.data(), sizeof(double) * h->D.P, hipMemcpyHostToDevice));
  });
}

int pmaf_peer_disconnect(pmaf_planner *h) {
  return guarded([&] {
    REQUIRE(h, "pmaf_peer_disconnect: NULL handle");
    h->use_device();
    sync(h);
    peer_disconnect(h);
  });
}

int pmaf_peer_read(pmaf_planner *h, double *headers, double *seq) {
  return guarded([&] {
    REQUIRE(h, "pmaf_peer_read: NULL handle");
    pmaf_planner::Peer &pr = h->peer;
    if (!pr.on) fail(PMAF_ERR_STATE, "pmaf_peer_read: no peer mailboxes connected (pmaf_peer_connect)");
    h->use_device();
    const int P = h->D.P;
    std::vector<double> box(peer_inbox_doubles(pr.world, P));
    HIP_CHECK(hipMemcpy(box.data(), pr.inbox, sizeof(double) * box.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < pr.world; r++)
      for (int p = 0; p < P; p++) {
        const double *s0 = box.data() + (((size_t)0 * pr.world + r) * P + p) * PMAF_PEER_SLOT;
        const double *s1 = box.data() + (((size_t)1 * pr.world + r) * P + p) * PMAF_PEER_SLOT;
        const double *s = s1[8] > s0[8] ? s1 : s0;   // the newer of the two parity slots
        if (headers) std::memcpy(headers + ((size_t)r * P + p) * PMAF_WINNER_HDR, s, sizeof(double) * PMAF_WINNER_HDR);
        if (seq) seq[(size_t)r * P + p] = s[8];
      }
  });
}

int pmaf_get_peer_times_us(pmaf_planner *h, double *wait_us, double *publish_us, int32_t max_n, int32_t *n) {
  return guarded([&] {
    REQUIRE(h && n, "pmaf_get_peer_times_us: NULL argument");
    pmaf_planner::Peer &pr = h->peer;
    const size_t k = max_n > 0 ? std::min(pr.wait_us.size(), (size_t)max_n) : 0;
    for (size_t i = 0; i < k; i++) {
      if (wait_us) wait_us[i] = pr.wait_us[i];
      if (publish_us) publish_us[i] = pr.pub_us[i];
    }
    *n = (int32_t)k;
    pr.wait_us.clear();
    pr.pub_us.clear();
  });
}

int pmaf_peer_info(pmaf_planner *h, int32_t *fine_grained, int32_t *world) {
  return guarded([&] {
    REQUIRE(h, "pmaf_peer_info: NULL handle");
    if (!h->peer.inbox) fail(PMAF_ERR_STATE, "pmaf_peer_info: no inbox (pmaf_peer_export)");
    if (fine_grained) *fine_grained = h->peer.inbox_fine ? 1 : 0;
    if (world) *world = h->peer.world;
  });
}

// ---- failure detection / winner path (ABI 5) ----
int pmaf_get_health(pmaf_planner *h, int32_t *bits) {
  return guarded([&] {
    REQUIRE(h && bits, "pmaf_get_health: NULL argument");
    for (int p = 0; p < h->D.P; p++) bits[p] = (int32_t)h->h_out[p * PMAF_MBOX + 15];   // (mailbox: host memory)
  });
}

int pmaf_debug_withhold_mailbox(pmaf_planner *h, int32_t enable) {
  return guarded([&] {
    REQUIRE(h, "pmaf_debug_withhold_mailbox: NULL handle");
    h->dbg_withhold = enable != 0;
  });
}

int pmaf_enable_winner_path(pmaf_planner *h, int32_t enable) {
  return guarded([&] {
    REQUIRE(h, "pmaf_enable_winner_path: NULL handle");
    h->use_device();
    sync(h);
    if (!enable) {
      if (h->h_wp) { (void)hipHostFree(h->h_wp); h->h_wp = nullptr; h->d_wp = nullptr; }
      if (h->h_wph) { (void)hipHostFree(h->h_wph); h->h_wph = nullptr; h->d_wph = nullptr; }
      h->wp_seq = 0.0;
      return;
    }
    if (h->h_wp) return;
    const size_t P = (size_t)h->D.P;
    HIP_CHECK(hipHostMalloc((void **)&h->h_wp, sizeof(double) * P * h->D.cap * 3, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h->d_wp, h->h_wp, 0));
    HIP_CHECK(hipHostMalloc((void **)&h->h_wph, sizeof(double) * P * 4, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h->d_wph, h->h_wph, 0));
    std::memset(h->h_wp, 0, sizeof(double) * P * h->D.cap * 3);
    std::memset(h->h_wph, 0, sizeof(double) * P * 4);
    h->wp_np.assign(P, 0);
    h->wp_agent.assign(P, 0);
    h->wp_seq = 0.0;
    h->wp_timed = true;
  });
}

int pmaf_view_winner_path(pmaf_planner *h, const double **paths, const int32_t **n_points, const int32_t **agent) {
  return guarded([&] {
    REQUIRE(h, "pmaf_view_winner_path: NULL handle");
    if (!h->h_wp) fail(PMAF_ERR_STATE, "pmaf_view_winner_path: not enabled (pmaf_enable_winner_path)");
    if (h->wp_seq == 0.0) fail(PMAF_ERR_STATE, "pmaf_view_winner_path: no selection yet (pmaf_tick / pmaf_evaluate), or the last "
                                               "tick's sequence number was withheld");
    // the path follows the set-point within microseconds (same kernel); bounded like the tick's own wait
    const auto t0 = std::chrono::steady_clock::now();
    for (int p = 0; p < h->D.P; p++) {
      const volatile double *s = h->h_wph + p * 4 + 3;
      unsigned spins = 0;
      while (*s != h->wp_seq) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((++spins & 0x3fffu) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > h->tick_timeout_s)
          fail(PMAF_ERR_DEVICE, "pmaf_view_winner_path: the path did not arrive within the time limit (PMAF_TICK_TIMEOUT_S)");
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (!h->wp_timed) {
      h->wp_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h->last_tick_entry).count());
      if (h->wp_us.size() > (1u << 20)) h->wp_us.erase(h->wp_us.begin(), h->wp_us.begin() + (1u << 19));
      h->wp_timed = true;
    }
    for (int p = 0; p < h->D.P; p++) {
      h->wp_np[p] = (int32_t)h->h_wph[p * 4];
      h->wp_agent[p] = (int32_t)h->h_wph[p * 4 + 1];
    }
    if (paths) *paths = h->h_wp;
    if (n_points) *n_points = h->wp_np.data();
    if (agent) *agent = h->wp_agent.data();
  });
}

int pmaf_get_winner_path_times_us(pmaf_planner *h, double *out, int32_t max_n, int32_t *n) {
  return guarded([&] {
    REQUIRE(h && n, "pmaf_get_winner_path_times_us: NULL argument");
    std::vector<double> &v = h->wp_us;
    const size_t k = (out && max_n > 0) ? std::min(v.size(), (size_t)max_n) : 0;
    for (size_t i = 0; i < k; i++) out[i] = v[i];
    *n = (int32_t)k;
    v.clear();
  });
}

int pmaf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

// ---- checkpoint / resume ----------------------------------------------------
// The blob holds every device buffer of the handle (agents' rotation vectors,
// known flags, paths, real agent, best-agent copy, obstacle tables ...) plus
// the host-side planner state (real agent's trajectory, scoring parameters).
// the blob stores the handle's own buffers: with a communicator attached the current paths may live in the second
// path buffer -- finish the exchange in flight and move them back
static void normalise_path_buffer(pmaf_planner *h) {
  if (!h->x.c) return;
  finish_exchanges(h);
  if (h->D.paths != h->x.paths_a) {
    HIP_CHECK(hipMemcpy(h->x.paths_a, h->x.paths_b, sizeof(double) * (size_t)h->D.P * h->D.N * h->D.cap * 3, hipMemcpyDeviceToDevice));
    h->D.paths = h->x.paths_a;
  }
}

struct StateHeader {
  uint64_t magic;
  int32_t abi, P, N, n_obs, cap, n_bufs;
  int32_t cp_valid, scores_valid, rollout_pending, stepped;
  uint64_t dev_bytes;
};
static const uint64_t kStateMagic = 0x504d41465f535431ull;  // "PMAF_ST1"

static size_t state_bytes(const pmaf_planner *h) {
  size_t n = sizeof(StateHeader) + sizeof(CostParams) + sizeof(PopConst);
  for (size_t b : h->alloc_bytes) n += b;
  n += sizeof(double) * 9 * (size_t)h->D.P;                    // host mirror of the real agent
  for (auto &rp : h->real_path) n += sizeof(uint64_t) + sizeof(double) * rp.size();
  return n;
}

size_t pmaf_state_size(const pmaf_planner *h) { return h ? state_bytes(h) : 0; }

int pmaf_save_state(pmaf_planner *h, void *blob, size_t bytes) {
  return guarded([&] {
    REQUIRE(h && blob, "pmaf_save_state: NULL argument");
    REQUIRE(bytes >= state_bytes(h), "pmaf_save_state: buffer too small (see pmaf_state_size)");
    h->use_device();
    sync(h);
    normalise_path_buffer(h);
    flush_real_position(h);   // (closed loop: a measured position not yet consumed by a manager launch)
    char *w = static_cast<char *>(blob);
    StateHeader hd{};
    hd.magic = kStateMagic; hd.abi = PMAF_ABI_VERSION;
    hd.P = h->D.P; hd.N = h->D.N; hd.n_obs = h->D.n_obs; hd.cap = h->D.cap; hd.n_bufs = (int32_t)h->allocs.size();
    hd.cp_valid = h->cp_valid; hd.scores_valid = h->scores_valid; hd.rollout_pending = h->rollout_pending;
    hd.stepped = h->stepped;
    hd.dev_bytes = 0;
    for (size_t b : h->alloc_bytes) hd.dev_bytes += b;
    std::memcpy(w, &hd, sizeof(hd)); w += sizeof(hd);
    std::memcpy(w, &h->cp, sizeof(CostParams)); w += sizeof(CostParams);
    std::memcpy(w, &h->D.C, sizeof(PopConst)); w += sizeof(PopConst);
    for (size_t i = 0; i < h->allocs.size(); i++) {
      HIP_CHECK(hipMemcpyAsync(w, h->allocs[i], h->alloc_bytes[i], hipMemcpyDeviceToHost, h->stream));
      w += h->alloc_bytes[i];
    }
    HIP_CHECK(hipStreamSynchronize(h->stream));
    const size_t n3 = sizeof(double) * 3 * (size_t)h->D.P;
    std::memcpy(w, h->real_pos_h.data(), n3); w += n3;
    std::memcpy(w, h->real_vel_h.data(), n3); w += n3;
    std::memcpy(w, h->real_force_h.data(), n3); w += n3;
    for (auto &rp : h->real_path) {
      uint64_t n = rp.size();
      std::memcpy(w, &n, sizeof(n)); w += sizeof(n);
      std::memcpy(w, rp.data(), sizeof(double) * n); w += sizeof(double) * n;
    }
  });
}

int pmaf_load_state(pmaf_planner *h, const void *blob, size_t bytes) {
  return guarded([&] {
    REQUIRE(h && blob, "pmaf_load_state: NULL argument");
    REQUIRE(bytes >= sizeof(StateHeader) + sizeof(CostParams) + sizeof(PopConst), "pmaf_load_state: blob too small");
    const char *r = static_cast<const char *>(blob);
    StateHeader hd;
    std::memcpy(&hd, r, sizeof(hd)); r += sizeof(hd);
    uint64_t dev = 0;
    for (size_t b : h->alloc_bytes) dev += b;
    REQUIRE(hd.magic == kStateMagic && hd.abi == PMAF_ABI_VERSION, "pmaf_load_state: not a pmaf state blob of this ABI version");
    REQUIRE(hd.P == h->D.P && hd.N == h->D.N && hd.n_obs == h->D.n_obs && hd.cap == h->D.cap &&
                hd.n_bufs == (int32_t)h->allocs.size() && hd.dev_bytes == dev,
            "pmaf_load_state: blob was saved from a handle with different dimensions");
    REQUIRE(bytes >= sizeof(StateHeader) + sizeof(CostParams) + sizeof(PopConst) + dev + sizeof(double) * 9 * (size_t)h->D.P,
            "pmaf_load_state: blob truncated");
    h->use_device();
    sync(h);
    normalise_path_buffer(h);
    std::memcpy(&h->cp, r, sizeof(CostParams)); r += sizeof(CostParams);
    std::memcpy(&h->D.C, r, sizeof(PopConst)); r += sizeof(PopConst);
    for (size_t i = 0; i < h->allocs.size(); i++) {
      HIP_CHECK(hipMemcpyAsync(h->allocs[i], r, h->alloc_bytes[i], hipMemcpyHostToDevice, h->stream));
      r += h->alloc_bytes[i];
    }
    HIP_CHECK(hipStreamSynchronize(h->stream));
    refresh_plain_step(h, nullptr);                                 // the blob's gains and mass, not the handle's
    h->download(h->goal_h.data(), h->D.goal, (size_t)h->D.P * 3);  // host copy of the goals
    const size_t n3 = sizeof(double) * 3 * (size_t)h->D.P;
    std::memcpy(h->real_pos_h.data(), r, n3); r += n3;
    std::memcpy(h->real_vel_h.data(), r, n3); r += n3;
    std::memcpy(h->real_force_h.data(), r, n3); r += n3;
    const char *end = static_cast<const char *>(blob) + bytes;
    for (auto &rp : h->real_path) {
      uint64_t n = 0;
      REQUIRE(r + sizeof(n) <= end, "pmaf_load_state: blob truncated");
      std::memcpy(&n, r, sizeof(n)); r += sizeof(n);
      REQUIRE(r + sizeof(double) * n <= end, "pmaf_load_state: blob truncated");
      rp.assign(reinterpret_cast<const double *>(r), reinterpret_cast<const double *>(r) + n);
      r += sizeof(double) * n;
    }
    h->paths_gen++;
    h->cp_valid = hd.cp_valid != 0;
    h->scores_valid = hd.scores_valid != 0;
    h->rollout_pending = hd.rollout_pending != 0;
    h->stepped = hd.stepped != 0;
    h->real_pos_pending = false;
    h->has_best_h = -1;        // (the blob's: read back when pmaf_move_real asks)
    h->closest_dirty = true;   // (the blob's table matches its obs_start; recompute at the next reset all the same)
    h->last_live.clear();
    h->live_resident.clear();
  });
}

void *pmaf_stream(pmaf_planner *h) { return h ? (void *)h->stream : nullptr; }

int pmaf_set_profiling(pmaf_planner *h, int32_t enable) {
  return guarded([&] {
    REQUIRE(h, "pmaf_set_profiling: NULL handle");
    h->use_device();
    sync(h);
    h->profiling = enable != 0;
    h->prof_every = enable > 1 ? enable : 1;
    h->prof_phase = 0;
  });
}
int pmaf_get_kernel_stats(pmaf_planner *h, double *rollout_ms, int64_t *launches, int64_t *agent_steps) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_kernel_stats")
    if (rollout_ms) *rollout_ms = h->rollout_ms;
    if (launches) *launches = h->profiling ? h->timed_launches : h->launches;
    // (all launches since the reset, timed or not: pmaf_get_launch_count)
    if (agent_steps) {
      unsigned long long s = 0;
      h->download(&s, D.step_counter, 1);
      *agent_steps = (int64_t)s;
    }
  });
}
int pmaf_get_launch_count(pmaf_planner *h, int64_t *launches) {
  return guarded([&] {
    REQUIRE(h && launches, "pmaf_get_launch_count: NULL argument");
    *launches = h->launches;
  });
}
int pmaf_reset_kernel_stats(pmaf_planner *h) {
  return guarded([&] {
    REQUIRE(h, "pmaf_reset_kernel_stats: NULL handle");
    h->use_device();
    sync(h);
    h->rollout_ms = 0.0;
    h->launches = 0;
    h->timed_launches = 0;
    HIP_CHECK(hipMemsetAsync(h->D.step_counter, 0, sizeof(unsigned long long), h->stream));
    sync(h);
  });
}
int pmaf_debug_math(int32_t op, int32_t n, const double *a, const double *b, double *out) {
  return guarded([&] {
    REQUIRE(a && b && out && n > 0 && op >= 0 && op <= 12, "pmaf_debug_math: bad argument");
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_CHECK(hipMalloc((void **)&da, sizeof(double) * n));
    HIP_CHECK(hipMalloc((void **)&db, sizeof(double) * n));
    HIP_CHECK(hipMalloc((void **)&dout, sizeof(double) * n));
    HIP_CHECK(hipMemcpy(da, a, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(db, b, sizeof(double) * n, hipMemcpyHostToDevice));
    pmaf_k_launch_debug_math(op, n, da, db, dout, nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  });
}

int pmaf_debug_external_rollout(pmaf_planner *h, const char *code_object_path, const char *kernel_name) {
  return guarded([&] {
    REQUIRE(h, "pmaf_debug_external_rollout: NULL handle");
    h->use_device();
    sync(h);
    if (h->ext_mod) { (void)hipModuleUnload(h->ext_mod); h->ext_mod = nullptr; h->ext_fn = nullptr; }
    if (!code_object_path) return;
    REQUIRE(kernel_name, "pmaf_debug_external_rollout: kernel name required");
    REQUIRE(h->lpa == 64, "pmaf_debug_external_rollout: wave-per-agent launches only (grid N x P, one wave per block)");
    HIP_CHECK(hipModuleLoad(&h->ext_mod, code_object_path));
    HIP_CHECK(hipModuleGetFunction(&h->ext_fn, h->ext_mod, kernel_name));
  });
}

int pmaf_get_launch_config(pmaf_planner *h, int32_t *lanes_per_agent, int32_t *n_blocks, int32_t *lds_bytes) {
  return guarded([&] {
    REQUIRE(h, "pmaf_get_launch_config: NULL handle");
    if (lanes_per_agent) *lanes_per_agent = h->lpa;
    if (n_blocks) *n_blocks = h->n_blocks * h->D.P;
    if (lds_bytes) *lds_bytes = (int32_t)h->lds_rollout;
  });
}

int pmaf_get_waves_per_agent(pmaf_planner *h, int32_t *waves_per_agent, int32_t *obstacles_per_wave) {
  return guarded([&] {
    REQUIRE(h, "pmaf_get_waves_per_agent: NULL handle");
    if (waves_per_agent) *waves_per_agent = h->mw_waves ? h->mw_waves : 1;
    if (obstacles_per_wave) *obstacles_per_wave = h->mw_waves ? h->mw_per : h->D.n_obs - 1;
  });
}

int pmaf_get_priority_slices(pmaf_planner *h, int32_t *enabled, int32_t *slice_ticks, int32_t *younger_of_8) {
  return guarded([&] {
    REQUIRE(h, "pmaf_get_priority_slices: NULL handle");
    const bool on = w64_sliced(h);
    if (enabled) *enabled = on ? 1 : 0;
    if (slice_ticks) *slice_ticks = 1 << h->D.prio_slice_log2;
    if (younger_of_8) *younger_of_8 = h->D.prio_younger_of_8;
  });
}

int32_t pmaf_pick_lanes_per_agent(int32_t n_agents, int32_t n_populations, int32_t n_field_obstacles, int32_t n_simds) {
  if (n_agents < 1 || n_populations < 1 || n_field_obstacles < 0) return 0;
  return pick_lpa(n_agents, n_populations, n_field_obstacles, n_simds > 0 ? n_simds : 1024);
}

double pmaf_estimate_rollout_us(int32_t lanes_per_agent, int32_t n_agents, int32_t n_populations, int32_t n_field_obstacles,
                                int32_t horizon, int32_t n_simds) {
  return pmaf_lpa::estimate_us(lanes_per_agent, n_agents, n_populations, n_field_obstacles, horizon, n_simds > 0 ? n_simds : 1024);
}

}  // extern "C"


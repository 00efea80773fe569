// pmaf_k_misc.hip -- the remaining kernels of libpmaf_hip.so and the launch interface of pmaf_types.hpp:
//   k_rollout<LPA>  generic agent x horizon rollout (CfAgent::cfPrediction, B/src/cf_agent.cpp:302-341): any
//                   lanes-per-agent mapping, obstacle table in LDS advanced once per step (fallback shape).
//   k_manager       one wave per population: evaluateAgents' cost assembly + argmin + hysteresis
//                   (B/src/cf_manager.cpp:325-353), RealCfAgent::cfPlanner single step (B/src/cf_agent.cpp:343-366)
//                   and resetEEAgents (B/src/cf_manager.cpp:246-255).
//   k_score         re-scores stored paths (before the first rollout / other workspace gains).
//   k_link_force    CfAgent::bodyForce (B/src/cf_agent.cpp:229-234).
//   k_winner        packs winner records for sharded runs.
#include <mutex>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "pmaf_types.hpp"
#include "pmaf_device.hpp"
#include "pmaf_rollout_w64.hpp"
#include "pmaf_rollout_grp.hpp"

using namespace pmaf;

// ---------------------------------------------------------------------------
// k_rollout
// ---------------------------------------------------------------------------
template <int LPA>
__global__ __launch_bounds__(64) void k_rollout(DevView D, CostParams CP) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const int pop = blockIdx.y;
  constexpr int APW = 64 / LPA;  // agents per wave
  const int sub = lane % LPA;
  const int grp = lane / LPA;
  const int a = blockIdx.x * APW + grp;
  const bool active = a < D.N;
  const int aa = active ? a : 0;
  const int n_obs = D.n_obs;
  const PopConst C = D.C;
  const unsigned long long t_begin = wall_clock64();

  ObsTab T = carve_obstab(smem, n_obs);
  int32_t *s_known = reinterpret_cast<int32_t *>(smem + 7 * n_obs);
  {
    const double *src = D.obs_start + (size_t)pop * 7 * n_obs;
    for (int i = lane; i < 7 * n_obs; i += 64) smem[i] = src[i];
    const int32_t *ks = D.known_start + (size_t)pop * n_obs;
    for (int i = lane; i < n_obs; i += 64) s_known[i] = ks[i];
  }
  __syncthreads();

  const size_t pa = (size_t)pop * D.N + aa;
  const V3 goal = mk(D.goal[pop * 3], D.goal[pop * 3 + 1], D.goal[pop * 3 + 2]);
  const V3 init_pos = mk(D.agent_init_pos[pop * 3], D.agent_init_pos[pop * 3 + 1], D.agent_init_pos[pop * 3 + 2]);
  V3 p = mk(D.start_pos[pop * 3], D.start_pos[pop * 3 + 1], D.start_pos[pop * 3 + 2]);
  V3 v = mk(D.start_vel[pop * 3], D.start_vel[pop * 3 + 1], D.start_vel[pop * 3 + 2]);
  const double k_attr = D.k_attr[pa], k_circ = D.k_circ[pa], k_repel = D.k_repel[pa], k_damp = D.k_damp[pa];
  const int type = D.types[aa];
  double *rot_g = D.rot + pa * 3 * n_obs;
  const double *rnd_g = D.rnd + pa * 3 * n_obs;
  double *path = D.paths + pa * (size_t)D.cap * 3;

  const int M = n_obs - 1;
  const int ntiles = (M + LPA - 1) / LPA;
  unsigned long long known_bits = 0ull;
  for (int t = 0; t < ntiles; t++) {
    int i = t * LPA + sub;
    if (i < M && s_known[i]) known_bits |= (1ull << t);
  }

  double min_obs = C.shell;
  double cost_ws = 0.0;
  double path_len = 0.0;
  int n = 1;
  bool ran = false;
  ws_cost_add(cost_ws, p, CP.ws, CP.k_workspace);
  if (active && sub == 0) { path[0] = p.x; path[1] = p.y; path[2] = p.z; }

  while (true) {
    V3 g = goal - p;
    double dg = norm(g);
    bool run = active && (dg > 0.1) && (n < D.cap);
    if (!__any(run)) break;
    // gate, B/src/cf_agent.cpp:315-317
    bool gate = !(dg < C.approach || (norm(v) < 0.5 * C.vel_max && norm(p - init_pos) < 0.2));
    V3 F = mk(0.0, 0.0, 0.0);
    double scale = 1.0;
    circ_and_scale<LPA, false>(run && gate, sub, grp, type, p, v, goal, g, C, k_circ, T, n_obs, rot_g,
                               rnd_g, known_bits, min_obs, F, scale);
    V3 new_pos;
    V3 nv = v;
    finish_step(p, nv, g, F, scale, C, k_attr, k_repel, k_damp, C.dt, T.pos(n_obs - 1), T.r[n_obs - 1], new_pos);
    if (run) {
      path_len += norm(new_pos - p);
      p = new_pos;
      v = nv;
      ws_cost_add(cost_ws, p, CP.ws, CP.k_workspace);
      PMAF_BOUND(n < D.cap);
      if (sub == 0) { path[n * 3] = p.x; path[n * 3 + 1] = p.y; path[n * 3 + 2] = p.z; }
      n++;
      ran = true;
    }
    // predictObstacles, B/src/cf_agent.cpp:270-276 (shared copy, once per step)
    __syncthreads();
    for (int i = lane; i < n_obs; i += 64) {
      T.px[i] = T.px[i] + T.vx[i] * C.dt;
      T.py[i] = T.py[i] + T.vy[i] * C.dt;
      T.pz[i] = T.pz[i] + T.vz[i] * C.dt;
    }
    __syncthreads();
  }

  if (active) {
    // known_obstacles_ of this agent after the rollout (getter only)
    int32_t *ko = D.known_out + pa * n_obs;
    for (int t = 0; t < ntiles; t++) {
      int i = t * LPA + sub;
      if (i < M) ko[i] = (int32_t)((known_bits >> t) & 1ull);
    }
    if (sub == 0) {
      ko[M] = s_known[M];
      D.n_points[pa] = n;
      D.agent_vel[pa * 3] = v.x; D.agent_vel[pa * 3 + 1] = v.y; D.agent_vel[pa * 3 + 2] = v.z;
      D.min_obs[pa] = min_obs;
      D.cost_ws[pa] = cost_ws;
      D.path_len[pa] = path_len;
      double dgf = norm(goal - p);
      D.goal_dist[pa] = dgf;
      if (ran) D.reached[pa] = dgf < 0.100001;  // B/src/cf_agent.cpp:330-337
      atomicAdd(D.step_counter, (unsigned long long)(n - 1));
      D.pred_ticks[pa] = wall_clock64() - t_begin;
    }
  }
}

// ---------------------------------------------------------------------------
// k_score: cost terms from stored paths (one thread per agent)
// ---------------------------------------------------------------------------
__global__ void k_score(DevView D, CostParams CP) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.P * D.N) return;
  int pop = idx / D.N;
  const double *path = D.paths + (size_t)idx * D.cap * 3;
  int n = D.n_points[idx];
  double cost = 0.0, len = 0.0;
  V3 prev = mk(path[0], path[1], path[2]);
  ws_cost_add(cost, prev, CP.ws, CP.k_workspace);
  for (int k = 1; k < n; k++) {
    V3 q = mk(path[k * 3], path[k * 3 + 1], path[k * 3 + 2]);
    ws_cost_add(cost, q, CP.ws, CP.k_workspace);
    len += norm(q - prev);
    prev = q;
  }
  V3 goal = mk(D.goal[pop * 3], D.goal[pop * 3 + 1], D.goal[pop * 3 + 2]);
  D.cost_ws[idx] = cost;
  D.path_len[idx] = len;
  D.goal_dist[idx] = norm(goal - prev);
}

// ---------------------------------------------------------------------------
// k_manager: evaluate / real step / reset, one wave per population
// ---------------------------------------------------------------------------
// RealCfAgent::cfPlanner, ONE step (B/src/cf_agent.cpp:343-366), with the tuned
// wave-per-agent step functions: the live obstacles, the real agent's rotation
// vectors and known flags sit in this wave's registers exactly as an agent's do
// in k_rollout_w64 (the generic LDS-table path took 7 us per call, this one
// is the same code the rollout spends ~2 us per step in). No min_obs_dist_
// tracking (RealCfAgent::circForce :110-144), heuristic = the stored best agent's.
template <int TILES>
__device__ __forceinline__ void real_step_w64(const DevView &D, const double dt_real, const int pop, const int lane,
                                              const int htype, const double *live, const int32_t *s_known,
                                              double *rot_g, const double *rand_g, double *clist, const double k_attr,
                                              const double k_circ, const double k_repel, const double k_damp,
                                              const V3 goal, const V3 init_pos, V3 &rp, V3 &rv, V3 &F_total,
                                              unsigned &known_bits, bool &acc_clamped) {
  typedef Mth<MATH_XACT> MT;
  PopConst C = D.C;
  C.dt = dt_real;
  const int n_obs = D.n_obs, M = n_obs - 1;
  LaneObstacles<TILES> O;
  known_bits = 0u;
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    const int i = t * 64 + lane;
    const bool valid = i < M;
    const int ii = valid ? i : 0;
    O.p[t] = mk(live[ii], live[n_obs + ii], live[2 * n_obs + ii]);
    O.v[t] = mk(live[3 * n_obs + ii], live[4 * n_obs + ii], live[5 * n_obs + ii]);
    O.r[t] = live[6 * n_obs + ii];
    O.rx[t] = rot_g[ii]; O.ry[t] = rot_g[n_obs + ii]; O.rz[t] = rot_g[2 * n_obs + ii];
    O.qx[t] = rand_g[ii]; O.qy[t] = rand_g[n_obs + ii]; O.qz[t] = rand_g[2 * n_obs + ii];
    if (valid && s_known[ii]) known_bits |= (1u << t);
  }
  const V3 sent_p = mk(live[M], live[n_obs + M], live[2 * n_obs + M]);
  const double sent_r = live[6 * n_obs + M];
  const V3 g = goal - rp;
  const double dg = MT::norm(g);
  const double zv = sqn(rv);
  const double z_init = sqn(rp - init_pos);
  const bool gate = !(dg < C.approach || (zv < C.zvhalf_lt && z_init < C.zinit_lt));  // :347-349
  const V3 gn = (dg > 0.0) ? MT::div3(g, dg) : g;
  const V3 verr = attractor_velocity_error<MATH_XACT>(rv, g, C, k_attr, k_damp);
  const V3 repel = sentinel_repel_live(rp, C, k_repel, sent_p, sent_r);  // live radius (the caller's list)
  V3 F = mk(0.0, 0.0, 0.0);
  double scale = 1.0, no_min = C.shell;
  SecTimers ST;
  if (gate)
    circ_and_scale_w64<TILES, T_REAL, MATH_XACT>(lane, rp, rv, zv, goal, g, dg, gn, C, k_circ, n_obs, rot_g, known_bits,
                                                 O, clist, no_min, F, scale, ST, exp_consts(), 0, htype);
  F = F + (mk(0.0, 0.0, 0.0) + repel);
  if (k_attr != 0.0) F = F + (scale * k_damp) * verr;
  F_total = F;
  V3 acc = F;
  if (C.mass != 1.0) acc = F / C.mass;
  const double az = sqn(acc);
  acc_clamped = az >= C.zacc_gt;
  if (az >= C.zacc_gt) acc = acc * (13.0 / __builtin_sqrt(az));
  const V3 half = ((0.5 * acc) * C.dt) * C.dt;
  const V3 new_pos = (rp + half) + (rv * C.dt);
  const V3 nv = rv + acc * C.dt;
  double vn, rvn;
  MT::norm_rcp(nv, vn, rvn);
  const V3 cl = nv * MT::div_n(C.vel_max, vn, rvn);
  rv = (vn > C.vel_max) ? cl : nv;
  rp = new_pos;
}


// Latency matters here (the set-point reaches the host when this kernel is
// done): everything that does not depend on the selection is loaded up front
// (per-agent results, the real agent's state, the live obstacles), the values
// that lanes exchange (costs, known flags, the obstacle table) go through LDS,
// not global memory, and the host-visible outputs are written before the
// agents' reset stores. The dependent global round trips on the critical path
// are: results -> (selected agent's type and gains) -> rotation vectors of the
// obstacles inside the real agent's shell.
__global__ __launch_bounds__(64) void k_manager(DevView D, CostParams CP, ManagerArgs A) {
  extern __shared__ double smem[];
#ifdef PMAF_TICK_STAMPS
  const unsigned long long t_mgr0 = wall_clock64();
#endif
#ifdef PMAF_MGR_SECTIONS   // timing experiments only: where the manager step's time goes (10 ns ticks, printed by population 0)
  unsigned long long t_sec[8];
  int n_sec = 0;
#define PMAF_MSEC() do { __builtin_amdgcn_s_waitcnt(0); t_sec[n_sec++] = wall_clock64(); } while (0)
#else
#define PMAF_MSEC() do { } while (0)
#endif
  PMAF_MSEC();
  const int lane = threadIdx.x;
  const int pop = blockIdx.x;
  const int n_obs = D.n_obs;
  const int N = D.N;
  const int M = n_obs - 1;
  const PopConst C = D.C;
  // LDS: live obstacle table [7][n_obs] | known flags [n_obs] i32 | costs [N]
  ObsTab T = carve_obstab(smem, n_obs);
  int32_t *s_known = reinterpret_cast<int32_t *>(smem + 7 * n_obs);
  double *s_cost = smem + 7 * n_obs + (n_obs + 1) / 2;

  // ---- loads that depend on nothing ----
#ifndef PMAF_MGR_EARLY_RESULTS
#define PMAF_MGR_EARLY_RESULTS 1
#endif
  // the rollout's per-agent results (first four agents of every lane = all of them up to 256 agents): requested before
  // anything else, so that their round trip runs under the obstacle table's instead of behind it (round 5, device-side
  // duration of this kernel in back-to-back C2 ticks: 11.05 -> 10.5 us; the same for every agent's gains and type, to take
  // the selected agent's out of a lane's registers after the argmin, was measured SLOWER: 11.3 us)
  double cw_pre[4], gd_pre[4], pl_pre[4], mo_pre[4];
  if (PMAF_MGR_EARLY_RESULTS && A.do_select) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int a = lane + 64 * k;
      const size_t pa = (size_t)pop * N + (a < N ? a : 0);
      cw_pre[k] = D.cost_ws[pa]; gd_pre[k] = D.goal_dist[pa]; pl_pre[k] = D.path_len[pa]; mo_pre[k] = D.min_obs[pa];
    }
  }
  const V3 goal = mk(D.goal[pop * 3], D.goal[pop * 3 + 1], D.goal[pop * 3 + 2]);
  int best = D.best_idx[pop];
  const int had_best = D.has_best[pop];
  const int old_best_id = D.best_id[pop];
  int htype = D.best_type[pop];
  // (closed loop: the measured position handed over by pmaf_set_real_position since the last manager launch)
  const double *rp_src = A.real_pos_src ? A.real_pos_src : D.real_pos;
  V3 rp;
  if (A.real_pos_inline) rp = mk(A.real_pos_val[pop * 3], A.real_pos_val[pop * 3 + 1], A.real_pos_val[pop * 3 + 2]);   // (kernel arguments)
  else rp = mk(rp_src[pop * 3], rp_src[pop * 3 + 1], rp_src[pop * 3 + 2]);
  if ((A.real_pos_src || A.real_pos_inline) && !A.do_move && lane == 0) {   // no step in this launch: keep it for the launches that follow
    D.real_pos[pop * 3] = rp.x; D.real_pos[pop * 3 + 1] = rp.y; D.real_pos[pop * 3 + 2] = rp.z;
  }
  V3 rv = mk(D.real_vel[pop * 3], D.real_vel[pop * 3 + 1], D.real_vel[pop * 3 + 2]);
  V3 rf = mk(D.real_force[pop * 3], D.real_force[pop * 3 + 1], D.real_force[pop * 3 + 2]);
  const V3 init_pos = mk(D.real_init_pos[pop * 3], D.real_init_pos[pop * 3 + 1], D.real_init_pos[pop * 3 + 2]);
  int32_t *rk = D.real_known + (size_t)pop * n_obs;
  // live obstacles: read straight from the caller's pinned staging buffer when the tick brought new ones (1.8 KB
  // over PCIe costs less than a copy command in front of this kernel) and kept in D.obs_live for later calls
  double *live_dev = D.obs_live + (size_t)pop * 7 * n_obs;
  const double *live = A.live_src ? A.live_src + (size_t)pop * 7 * n_obs : live_dev;
  if (A.do_move || A.do_reset) {
    if (A.live_src) {
      // a NEW list (moving obstacles streamed at the tick rate): every load is a PCIe read of ~1 us, and a rolled loop
      // with the stores in between issued them one after the other (33 obstacles: +4.5 us of set-point latency, 129:
      // +17 us). Batches of eight loads per lane in flight together, stored afterwards: one round trip per 512 values
      // (round 5, tools/ticklat.py, set-point latency with a new list per tick: C2 17.3 -> 14.9 us, C3's 129 obstacles 33.1 -> 20.0 us;
      // static lists: 13.4 us).
      const int total = 7 * n_obs;
      for (int base = 0; base < total; base += 64 * 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int i = base + k * 64 + lane;
          v[k] = (i < total) ? __builtin_nontemporal_load(live + i) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int i = base + k * 64 + lane;
          if (i < total) { smem[i] = v[k]; live_dev[i] = v[k]; }
        }
      }
    } else {
      for (int i = lane; i < 7 * n_obs; i += 64) smem[i] = live[i];
    }
    for (int i = lane; i < n_obs; i += 64) s_known[i] = rk[i];
  }
  // ---- peer mailboxes: a coupled population's trailing repulsive obstacle = the source population's set-point of
  // the PREVIOUS tick, read from this rank's own inbox (the source rank's k_manager stored it there, over xGMI when
  // it is another GPU). The header of tick t-1 sits in parity slot (t-1) & 1; its writer cannot overwrite it before
  // this rank has published tick t (it needs that header for its tick t+1), which happens further down.
  unsigned long long peer_wait = 0ull;
  int peer_status = 0;
  const PeerView *PV = A.peer;
  const bool peer_on = PV != nullptr && A.do_select && A.do_move && A.do_reset;
  if (peer_on) {
    const int src_rank = PV->couple[pop * 2], src_pop = PV->couple[pop * 2 + 1];
    if (src_rank >= 0) {
      const double want = A.peer_tick - 1.0;
      const double *slot = PV->inbox + ((((size_t)((long long)want & 1) * PV->world + src_rank) * PV->P + src_pop) * PMAF_PEER_SLOT);
      const unsigned long long t0 = wall_clock64();
      // every lane polls the same address (one broadcast load); system scope: the writer is another agent
      for (;;) {
        const double got = __hip_atomic_load(slot + 8, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (got == want) break;
        // the slot already holds a NEWER header of the same parity: the source ran ahead and overwrote the one this
        // tick needs. With pairwise mutual couplings that cannot happen (the source's tick t+1 needs THIS rank's header
        // t first); one-way couplings and rings have no such back-pressure and are rejected (status 3 below, once the
        // source's first kernel-written header is seen) -- reported at once instead of after the time-out
        if (got > want) { peer_status = 2; break; }
        if (wall_clock64() - t0 > PV->timeout_ticks) { peer_status = 1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      peer_wait = wall_clock64() - t0;
      if (peer_status == 0) {
        // the publisher's own coupling rides in the header (slot[9], [10]; -2 in a host-written initial header): it must
        // point back at this population
        const double br = __hip_atomic_load(slot + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const double bp = __hip_atomic_load(slot + 10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (br != -2.0 && !(br == (double)PV->rank && bp == (double)pop)) peer_status = 3;
      }
      const double sx = __hip_atomic_load(slot + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const double sy = __hip_atomic_load(slot + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const double sz = __hip_atomic_load(slot + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const double sr = PV->couple_radius[pop];
      wave_lds_fence();  // the table rows other lanes stored above
      if (lane == 0 && peer_status == 0) {
        const int k = n_obs - 1;
        const double row[7] = {sx, sy, sz, 0.0, 0.0, 0.0, sr};
#pragma unroll
        for (int c = 0; c < 7; c++) { smem[c * n_obs + k] = row[c]; live_dev[c * n_obs + k] = row[c]; }
      }
    }
  }
  int health = 0;   // PMAF_HB_* bits of this tick (mailbox entry 15; wave-uniform)
  // random vectors the real agent's heuristic uses (best_agent_'s copy)
  const double *rand_g = D.best_rnd + (size_t)pop * 3 * n_obs;
  PMAF_MSEC();   // 1: up-front loads, live obstacles in LDS

  if (A.do_select) {
    // cost assembly + argmin, B/src/cf_manager.cpp:325-343
    double lmin = 1.7976931348623157e308;
    int lidx = 0x7fffffff;
    // (round 4: four agents per lane and pass with their sixteen loads issued together -- the stores to D.costs may alias
    // the result arrays as far as the compiler knows, so the rolled loop paid one memory round trip per agent: 7.4 us of
    // the 15 us manager kernel at 1024 agents; same operations per agent, ascending agent index per lane as before)
    for (int a0 = lane; a0 < N; a0 += 256) {
      double cw[4], gdv[4], pl[4], mov[4];
      if (PMAF_MGR_EARLY_RESULTS && a0 == lane) {
#pragma unroll
        for (int k = 0; k < 4; k++) { cw[k] = cw_pre[k]; gdv[k] = gd_pre[k]; pl[k] = pl_pre[k]; mov[k] = mo_pre[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int a = a0 + 64 * k;
          const size_t pa = (size_t)pop * N + (a < N ? a : a0);
          cw[k] = D.cost_ws[pa]; gdv[k] = D.goal_dist[pa]; pl[k] = D.path_len[pa]; mov[k] = D.min_obs[pa];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int a = a0 + 64 * k;
        if (a < N) {
          const size_t pa = (size_t)pop * N + a;
          double cost = cw[k];
          const double gd = gdv[k];
          if (gd > C.approach) cost += gd * CP.k_goal_dist;
          cost += pl[k] * CP.k_path_len;
          const double mo = mov[k];
          cost += CP.k_safe_dist / mo;
          if (mo < 2e-5) cost += 10000.0;
          D.costs[pa] = cost;
          s_cost[a] = cost;
          if (cost < lmin) { lmin = cost; lidx = a; }
        }
      }
    }
    group_argmin<64>(lmin, lidx);
    int min_idx = (lidx == 0x7fffffff) ? 0 : lidx;
    // no agent with a comparable cost (every cost NaN -- e.g. a NaN goal distance): the reference's argmin then keeps
    // index 0 (B/src/cf_manager.cpp:336-343); reported
    if (lidx == 0x7fffffff) health |= PMAF_HB_COST_NAN;
    wave_lds_fence();  // costs visible to the whole wave
    // hysteresis, :344-353
    bool take;
    if (had_best) {
      int bi = old_best_id - 1;
      double cb = s_cost[bi];
      double cm = s_cost[min_idx];
      if (cm < 0.9 * cb) take = true;
      else { take = false; min_idx = bi; }
    } else {
      take = true;
    }
    if (take) {  // best_agent_ = makeCopy()
      const double *src = D.rnd + ((size_t)pop * N + min_idx) * 3 * n_obs;
      double *dst = D.best_rnd + (size_t)pop * 3 * n_obs;
      for (int i = lane; i < 3 * n_obs; i += 64) dst[i] = src[i];
      rand_g = src;  // same values; the copy need not have landed
      htype = D.types[min_idx];
      if (lane == 0) {
        D.has_best[pop] = 1;
        D.best_id[pop] = min_idx + 1;
        D.best_type[pop] = htype;
      }
    }
    best = min_idx;
    if (lane == 0) D.best_idx[pop] = best;
  }
  PMAF_MSEC();   // 2: selection
  if (A.do_move) {
    // RealCfAgent::cfPlanner one step, B/src/cf_agent.cpp:343-366
    int gid = A.agent_id_inline ? A.agent_id_val[pop] : (A.agent_id ? A.agent_id[pop] : best);
    size_t pg = (size_t)pop * N + gid;
    double k_attr = D.k_attr[pg], k_circ = D.k_circ[pg], k_repel = D.k_repel[pg], k_damp = D.k_damp[pg];
    wave_lds_fence();  // obstacle table + known flags in LDS
    const int ntiles = (M + 63) / 64;
    unsigned long long kb = 0ull;
    V3 F = mk(0.0, 0.0, 0.0);
    bool acc_clamped = false;
    if (A.tuned_real_step && ntiles <= 4) {
      double *rrot = D.real_rot + (size_t)pop * 3 * n_obs;
      double *clist = s_cost + N + (N & 1);
      unsigned kb32 = 0u;
      if (ntiles <= 1)
        real_step_w64<1>(D, A.dt_real, pop, lane, htype, smem, s_known, rrot, rand_g, clist, k_attr, k_circ, k_repel,
                         k_damp, goal, init_pos, rp, rv, F, kb32, acc_clamped);
      else if (ntiles == 2)
        real_step_w64<2>(D, A.dt_real, pop, lane, htype, smem, s_known, rrot, rand_g, clist, k_attr, k_circ, k_repel,
                         k_damp, goal, init_pos, rp, rv, F, kb32, acc_clamped);
      else
        real_step_w64<4>(D, A.dt_real, pop, lane, htype, smem, s_known, rrot, rand_g, clist, k_attr, k_circ, k_repel,
                         k_damp, goal, init_pos, rp, rv, F, kb32, acc_clamped);
      kb = kb32;
    } else {
      for (int t = 0; t < ntiles; t++) {
        int i = t * 64 + lane;
        if (i < M && s_known[i]) kb |= (1ull << t);
      }
      V3 g = goal - rp;
      double dg = norm(g);
      bool gate = !(dg < C.approach || (norm(rv) < 0.5 * C.vel_max && norm(rp - init_pos) < 0.2));
      double scale = 1.0, dummy_min = C.shell;
      circ_and_scale<64, true>(gate, lane, 0, htype, rp, rv, goal, g, C, k_circ, T, n_obs,
                               D.real_rot + (size_t)pop * 3 * n_obs, rand_g, kb, dummy_min, F, scale);
      V3 new_pos;
      finish_step(rp, rv, g, F, scale, C, k_attr, k_repel, k_damp, A.dt_real, T.pos(n_obs - 1), T.r[n_obs - 1], new_pos);
      rp = new_pos;
      // (F now holds the total force; the clamp test of updatePositionAndVelocity :255-257 on it)
      acc_clamped = norm((C.mass != 1.0) ? (F / C.mass) : F) > 13.0;
    }
    if (acc_clamped) health |= PMAF_HB_ACC_CLAMPED;
    rf = F;
    for (int t = 0; t < ntiles; t++) {
      int i = t * 64 + lane;
      if (i < M) {
        const int32_t f = (int32_t)((kb >> t) & 1ull);
        rk[i] = f;
        s_known[i] = f;
      }
    }
    if (lane == 0) {
      D.real_pos[pop * 3] = rp.x; D.real_pos[pop * 3 + 1] = rp.y; D.real_pos[pop * 3 + 2] = rp.z;
      D.real_vel[pop * 3] = rv.x; D.real_vel[pop * 3 + 1] = rv.y; D.real_vel[pop * 3 + 2] = rv.z;
      D.real_force[pop * 3] = F.x; D.real_force[pop * 3 + 1] = F.y; D.real_force[pop * 3 + 2] = F.z;
    }
  }

  PMAF_MSEC();   // 3: the real agent's step
  // header of this population's winner record (sharded runs): the selection just made, with the path length the
  // selected agent's rollout had when it was scored and the set-point the real agent moves to. Written BEFORE the
  // mailbox's sequence number (system-scope fence below): once the host has seen that number the header is visible
  // device-wide, and the host may enqueue the pack kernel + all-gather on another stream with no event in between
  double hv[PMAF_WINNER_HDR];
  const bool want_hdr = A.do_select && (A.winner_hdr != nullptr || peer_on);
  if (want_hdr) {  // wave-uniform values
    const size_t pb = (size_t)pop * N + best;
    hv[0] = s_cost[best];
    hv[1] = (double)best;
    hv[2] = (double)D.n_points[pb];
    hv[3] = (double)D.types[best];
    hv[4] = rp.x; hv[5] = rp.y; hv[6] = rp.z;
    hv[7] = norm(goal - rp);
  }
  if (lane == 0 && A.winner_hdr && A.do_select) {
    double *w = A.winner_hdr + (size_t)pop * A.winner_stride;
#pragma unroll
    for (int c = 0; c < PMAF_WINNER_HDR; c++) w[c] = hv[c];
  }
  // peer mailboxes: lane l stores the header into rank l's inbox (its own included), slot [t & 1][rank][pop];
  // the sequence number follows behind the system-scope fence shared with the host mailbox below
  unsigned long long peer_pub0 = 0ull;
  double *peer_slot = nullptr;
  if (peer_on) {
    peer_pub0 = wall_clock64();
    if (lane < PV->world) {
      peer_slot = PV->peer[lane] + ((((size_t)((long long)A.peer_tick & 1) * PV->world + PV->rank) * PV->P + pop) * PMAF_PEER_SLOT);
#pragma unroll
      for (int c = 0; c < PMAF_WINNER_HDR; c++) __hip_atomic_store(peer_slot + c, hv[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // this population's own coupling, for the consumer's mutuality check above
      __hip_atomic_store(peer_slot + 9, (double)PV->couple[pop * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(peer_slot + 10, (double)PV->couple[pop * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }

  // pmaf_reset_agents with more than PMAF_RP_INLINE populations: position / velocity come from a device buffer. Loaded
  // HERE, in front of the publication: the host may reuse the buffer as soon as the mailbox is out (with the publication in
  // front of this load only the stream order of the next upload protected it).
  V3 reset_sp = mk(0.0, 0.0, 0.0), reset_sv = mk(0.0, 0.0, 0.0);
  if (A.do_reset && !A.reset_from_real && !A.reset_in_inline) {
    const double *in = A.reset_in + pop * 6;
    reset_sp = mk(in[0], in[1], in[2]);
    reset_sv = mk(in[3], in[4], in[5]);
  }
  // host-visible outputs first: the caller waits for these only
  if (A.out) {
    double *o = A.out + pop * PMAF_MBOX;
    if (lane == 0) {
      o[0] = (double)best;
      o[1] = rp.x; o[2] = rp.y; o[3] = rp.z;
      o[4] = rv.x; o[5] = rv.y; o[6] = rv.z;
      o[7] = norm(goal - rp);
      o[8] = rf.x; o[9] = rf.y; o[10] = rf.z;
      o[12] = (double)peer_wait;
      o[14] = (double)peer_status;
      // health of this tick: a NaN / infinite set-point (the reference's consumer only logs it, B/src/costp_controller.cpp:317-319),
      // a non-finite force on the real agent (Had's rotation vector is NaN by construction on the goal line,
      // B/src/cf_agent.cpp:599-611), the acceleration clamp hit, no comparable cost
      const double spsum = ((rp.x + rp.y) + rp.z) + ((rv.x + rv.y) + rv.z);
      const double fsum = (rf.x + rf.y) + rf.z;
      int hb = health;
      if (!(fabs(spsum) <= 1.7976931348623157e308)) hb |= PMAF_HB_SETPOINT_NAN;
      if (!(fabs(fsum) <= 1.7976931348623157e308)) hb |= PMAF_HB_FORCE_NAN;
      o[15] = (double)hb;
    }
    if (A.seq != 0.0 || peer_on) {
      __threadfence_system();  // entries 0..10 visible to the host (and the peers' headers to them) before the sequence numbers
      if (peer_slot) __hip_atomic_store(peer_slot + 8, A.peer_tick, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (lane == 0) {
        if (A.seq != 0.0) *reinterpret_cast<volatile double *>(o + 11) = A.seq;
        // statistics only, not ordered against the sequence number: the host may read the previous tick's value
        if (peer_on) o[13] = (double)(wall_clock64() - peer_pub0);
      }
    }
  }

  PMAF_MSEC();   // 4: published (mailbox, fence)
  // winner path (pmaf_enable_winner_path): the selected agent's path AS IT WAS SCORED, into mapped pinned host memory,
  // behind the set-point (which the caller waits for first) and before the rollout launched next overwrites the buffer
  // (stream order). 24 B per point over PCIe; its own sequence word last, behind a system-scope fence.
  if (A.wp_out && A.do_select) {
    const size_t pb = (size_t)pop * N + best;
    const int n3 = D.n_points[pb] * 3;
    const double *src = D.paths + pb * (size_t)D.cap * 3;
    double *dst = A.wp_out + (size_t)pop * D.cap * 3;
    for (int i = lane; i < n3; i += 64) dst[i] = src[i];
    double *wh = A.wp_hdr + pop * 4;
    if (lane == 0) { wh[0] = (double)(n3 / 3); wh[1] = (double)best; wh[2] = 0.0; }
    __threadfence_system();
    if (lane == 0) *reinterpret_cast<volatile double *>(wh + 3) = A.seq;
  }
  if (A.do_reset) {
    // resetEEAgents, B/src/cf_manager.cpp:246-255
    V3 sp, sv;
    if (A.reset_from_real) { sp = rp; sv = rv; }
    else if (A.reset_in_inline) {
      sp = mk(A.reset_in_val[pop * 6], A.reset_in_val[pop * 6 + 1], A.reset_in_val[pop * 6 + 2]);
      sv = mk(A.reset_in_val[pop * 6 + 3], A.reset_in_val[pop * 6 + 4], A.reset_in_val[pop * 6 + 5]);
    } else { sp = reset_sp; sv = reset_sv; }
    // setVelocity clamp, B/src/cf_agent.cpp:54-61
    double vn = norm(sv);
    if (vn > C.vel_max) sv = (C.vel_max / vn) * sv;
    wave_lds_fence();  // table / known flags (written above by other lanes)
    // setObstacles, :63-70: position and velocity from the live obstacles,
    // radius keeps its init value; known flags from the real agent
    double *st = D.obs_start + (size_t)pop * 7 * n_obs;
    for (int i = lane; i < 6 * n_obs; i += 64) st[i] = smem[i];
    int32_t *ks = D.known_start + (size_t)pop * n_obs;
    for (int i = lane; i < n_obs; i += 64) ks[i] = s_known[i];
    // The agents' own copies (path start, point count, velocity, min_obs_dist_, known_obstacles_) are what
    // getters see between a reset and the next rollout. When the rollout is launched right behind this
    // kernel (pmaf_tick) it rewrites every one of them, so the stores are skipped.
    if (!A.rollout_follows) {
      for (int a = lane; a < N; a += 64) {
        size_t pa = (size_t)pop * N + a;
        double *path = D.paths + pa * (size_t)D.cap * 3;
        path[0] = sp.x; path[1] = sp.y; path[2] = sp.z;
        D.n_points[pa] = 1;
        D.agent_vel[pa * 3] = sv.x; D.agent_vel[pa * 3 + 1] = sv.y; D.agent_vel[pa * 3 + 2] = sv.z;
        D.min_obs[pa] = C.shell;
      }
      // known_obstacles_ of every agent <- the real agent's flags (setObstacles, cf_agent.cpp:68):
      // one coalesced sweep over [N][n_obs] instead of a per-agent loop
      int32_t *ko = D.known_out + (size_t)pop * N * n_obs;
      for (int k = lane; k < N * n_obs; k += 64) ko[k] = s_known[k % n_obs];
    }
    if (lane == 0) {
      D.start_pos[pop * 3] = sp.x; D.start_pos[pop * 3 + 1] = sp.y; D.start_pos[pop * 3 + 2] = sp.z;
      D.start_vel[pop * 3] = sv.x; D.start_vel[pop * 3 + 1] = sv.y; D.start_vel[pop * 3 + 2] = sv.z;
    }
    // Closest-other table of the obstacle list just copied into obs_start (DevView::closest_idx): only when the caller
    // handed over new obstacles since the last one, behind everything the host and the next rollout wait for, and only
    // for field obstacles at rest -- then every rollout's Obstacle / GoalObstacle latches read the answer of the
    // reference's scan (closest_other, B/src/cf_agent.cpp:434-446) instead of searching (M distances per latch).
    if (A.compute_closest) {
      bool mv = false;
      for (int i = lane; i < M; i += 64) {
        const V3 ov = T.vel(i);
        mv = mv || !(ov.x == 0.0 && ov.y == 0.0 && ov.z == 0.0);   // (a NaN component counts as moving)
      }
      mv = wave_any(mv);
      if (!mv) {
        int32_t *ci = D.closest_idx + (size_t)pop * n_obs;
        for (int i = lane; i < M; i += 64) ci[i] = closest_other(T, n_obs, i, T.pos(i));
      }
      if (lane == 0) D.closest_ok[pop] = mv ? 0 : 1;
    }
  }
#ifdef PMAF_TICK_STAMPS
  if (lane == 0 && pop == 0) printf("M %llu %llu\n", t_mgr0, wall_clock64());
#endif
#ifdef PMAF_MGR_SECTIONS
  PMAF_MSEC();   // 5: reset
  if (lane == 0 && pop == 0)
    printf("MGR loads %llu select %llu real-step %llu publish %llu reset %llu | total %llu (x10 ns)\n", t_sec[1] - t_sec[0],
           t_sec[2] - t_sec[1], t_sec[3] - t_sec[2], t_sec[4] - t_sec[3], t_sec[5] - t_sec[4], t_sec[5] - t_sec[0]);
#endif
#undef PMAF_MSEC
}

// CfAgent::setPosition for every predicted agent (clear + push_back,
// B/src/cf_agent.cpp:39-42): 1-point paths at pos[pop]
__global__ void k_restart_paths(DevView D, const double *pos) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.P * D.N) return;
  int pop = idx / D.N;
  double *path = D.paths + (size_t)idx * D.cap * 3;
  path[0] = pos[pop * 3]; path[1] = pos[pop * 3 + 1]; path[2] = pos[pop * 3 + 2];
  D.n_points[idx] = 1;
}

// CfAgent::bodyForce -> repelForce, B/src/cf_agent.cpp:229-234, 159-181
__global__ void k_link_force(int n, const double *link_pos, const double *k_r, const double *sent /*7*/,
                             double rad, double shell, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 p = mk(link_pos[3 * i], link_pos[3 * i + 1], link_pos[3 * i + 2]);
  V3 sp = mk(sent[0], sent[1], sent[2]);
  V3 ro = sp - p;
  V3 dist_vec = -ro;
  double d = norm(dist_vec) - (rad + sent[6]);
  d = smax(d, 1e-5);
  V3 repel = mk(0.0, 0.0, 0.0);
  if (d < shell) {
    V3 otr = normalized(p - sp);
    double t = 1.0 / d - 1.0 / shell;
    double dd = d * d;
    repel = ((k_r[i] * otr) * t) / dd;
  }
  V3 F = mk(0.0, 0.0, 0.0) + (mk(0.0, 0.0, 0.0) + repel);
  out[3 * i] = F.x; out[3 * i + 1] = F.y; out[3 * i + 2] = F.z;
}

// elementary operations the parity argument rests on, exposed for the GPU
// self-test: 0 a/b, 1 sqrt(a), 2 exp(a), 3 a*b, 4 a+b (compiler sequences);
// 5 Xact::sqrt(a), 6 Xact::div(a,b), 7 / 8 a 3-vector divided by a scalar
// through Xact::div3 / the compiler (summed to one double)
__global__ void k_debug_math(int op, int n, const double *a, const double *b, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0.0;
  switch (op) {
    case 0: r = a[i] / b[i]; break;
    case 1: r = __builtin_sqrt(a[i]); break;
    case 2: r = portable_exp(a[i]); break;
    case 3: r = a[i] * b[i]; break;
    case 4: r = a[i] + b[i]; break;
    case 5: r = Mth<MATH_XACT>::sqrt(a[i]); break;
    case 6: r = Mth<MATH_XACT>::div(a[i], b[i]); break;
    case 7: { V3 q = Mth<MATH_XACT>::div3(mk(a[i], b[i], a[i] * 0.5), b[i] + a[i]); r = (q.x + q.y) + q.z; } break;
    case 8: { V3 q = mk(a[i], b[i], a[i] * 0.5) / (b[i] + a[i]); r = (q.x + q.y) + q.z; } break;
    case 9: { double sq = Mth<MATH_XACT>::sqrt(b[i]); r = Mth<MATH_XACT>::div_r(a[i], sq, Mth<MATH_XACT>::rcp_refined(sq)); } break;  // a / sqrt(b)
    case 10: r = a[i] / __builtin_sqrt(b[i]); break;
    case 11: r = Mth<MATH_XACT>::div_pos(a[i], b[i]); break;                      // fixup-free a / b, b > 0 normal
    case 12: { double sq = Mth<MATH_XACT>::sqrt(b[i]); r = Mth<MATH_XACT>::div_r_pos(a[i], sq, Mth<MATH_XACT>::rcp_refined(sq)); } break;  // a / sqrt(b)
  }
  out[i] = r;
}

// winner record per population: {cost, idx, n_points, type, real agent's position[3], its goal distance,
// path[cap][3]} = (PMAF_WINNER_HDR + 3 cap) doubles; path entries past n_points are zero
__global__ void k_winner(DevView D, double *dst) {
  int pop = blockIdx.x;
  int best = D.best_idx[pop];
  size_t pa = (size_t)pop * D.N + best;
  size_t rec = PMAF_WINNER_HDR + (size_t)D.cap * 3;
  double *o = dst + pop * rec;
  if (threadIdx.x == 0) {
    o[0] = D.costs[pa];
    o[1] = (double)best;
    o[2] = (double)D.n_points[pa];
    o[3] = (double)D.types[best];
    const V3 rp = mk(D.real_pos[pop * 3], D.real_pos[pop * 3 + 1], D.real_pos[pop * 3 + 2]);
    const V3 goal = mk(D.goal[pop * 3], D.goal[pop * 3 + 1], D.goal[pop * 3 + 2]);
    o[4] = rp.x; o[5] = rp.y; o[6] = rp.z;
    o[7] = norm(goal - rp);
  }
  const double *path = D.paths + pa * (size_t)D.cap * 3;
  int n3 = D.n_points[pa] * 3;
  for (int i = threadIdx.x; i < D.cap * 3; i += blockDim.x) o[PMAF_WINNER_HDR + i] = (i < n3) ? path[i] : 0.0;
}

// the path part only: header (incl. n_points) already written by k_manager; `paths` is the buffer the scored
// rollout wrote (the handle may already be rolling out into its other path buffer)
__global__ void k_winner_path(DevView D, const double *paths, double *dst) {
  int pop = blockIdx.x;
  size_t rec = PMAF_WINNER_HDR + (size_t)D.cap * 3;
  double *o = dst + pop * rec;
  const int best = (int)o[1];
  const int n3 = (int)o[2] * 3;
  const double *path = paths + ((size_t)pop * D.N + best) * (size_t)D.cap * 3;
  for (int i = threadIdx.x; i < D.cap * 3; i += blockDim.x) o[PMAF_WINNER_HDR + i] = (i < n3) ? path[i] : 0.0;
}

// ---------------------------------------------------------------------------
// synchronous stepping API (no callers in the reference, kept for the class surface)
// ---------------------------------------------------------------------------
// CfAgent::cfPlanner, B/src/cf_agent.cpp:278-300, for every agent (CfManager::moveAgents / moveAgentsPar,
// B/src/cf_manager.cpp:274-291) or one agent until it is within 0.05 of the goal (moveAgent, :265-272): `steps`
// steps from the agent's CURRENT state (last path point, its own velocity, known flags, rotation vectors,
// min_obs_dist_) through the caller's obstacle list -- no loop guard, no obstacle advance, the call's delta_t.
// Same lanes-per-agent mapping and device functions as the generic rollout kernel.
template <int LPA>
__global__ __launch_bounds__(64) void k_plan_steps(DevView D, PlanArgs A) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const int pop = blockIdx.y;
  constexpr int APW = 64 / LPA;
  const int sub = lane % LPA;
  const int grp = lane / LPA;
  const int a = blockIdx.x * APW + grp;
  const bool exists = a < D.N;
  const int aa = exists ? a : 0;
  const bool chosen = exists && (A.only == nullptr || A.only[pop] == a);
  const int n_obs = D.n_obs;
  const PopConst C = D.C;
  ObsTab T = carve_obstab(smem, n_obs);
  {
    const double *src = A.obs + (size_t)pop * 7 * n_obs;
    for (int i = lane; i < 7 * n_obs; i += 64) smem[i] = src[i];
  }
  __syncthreads();
  const size_t pa = (size_t)pop * D.N + aa;
  const V3 goal = mk(D.goal[pop * 3], D.goal[pop * 3 + 1], D.goal[pop * 3 + 2]);
  const V3 init_pos = mk(D.agent_init_pos[pop * 3], D.agent_init_pos[pop * 3 + 1], D.agent_init_pos[pop * 3 + 2]);
  const double k_attr = D.k_attr[pa], k_circ = D.k_circ[pa], k_repel = D.k_repel[pa], k_damp = D.k_damp[pa];
  const int type = D.types[aa];
  double *rot_g = D.rot + pa * 3 * n_obs;
  const double *rnd_g = D.rnd + pa * 3 * n_obs;
  double *path = D.paths + pa * (size_t)D.cap * 3;
  int32_t *ko = D.known_out + pa * n_obs;
  int n = D.n_points[pa];
  V3 p = mk(path[(n - 1) * 3], path[(n - 1) * 3 + 1], path[(n - 1) * 3 + 2]);
  V3 v = mk(D.agent_vel[pa * 3], D.agent_vel[pa * 3 + 1], D.agent_vel[pa * 3 + 2]);
  double min_obs = D.min_obs[pa];
  const int M = n_obs - 1;
  const int ntiles = (M + LPA - 1) / LPA;
  unsigned long long known_bits = 0ull;
  for (int t = 0; t < ntiles; t++) {
    int i = t * LPA + sub;
    if (i < M && ko[i]) known_bits |= (1ull << t);
  }
  int calls = 0;
  bool active = chosen;
  for (int call = 0; call < A.max_calls; call++) {
    // moveAgent: `while (run_prediction_ && getDistFromGoal() > 0.05)`, cf_manager.cpp:267; the path buffer bounds it
    if (A.until_goal) active = active && (norm(goal - p) > 0.05);
    active = active && (n + A.steps <= D.cap);
    if (!__any(active)) break;
    if (active) calls++;
    for (int s = 0; s < A.steps; s++) {
      const V3 g = goal - p;
      const double dg = norm(g);
      const bool gate = !(dg < C.approach || (norm(v) < 0.5 * C.vel_max && norm(p - init_pos) < 0.2));  // :285-289
      V3 F = mk(0.0, 0.0, 0.0);
      double scale = 1.0;
      circ_and_scale<LPA, false>(active && gate, sub, grp, type, p, v, goal, g, C, k_circ, T, n_obs, rot_g, rnd_g,
                                 known_bits, min_obs, F, scale);
      V3 new_pos;
      V3 nv = v;
      finish_step(p, nv, g, F, scale, C, k_attr, k_repel, k_damp, A.dt, T.pos(n_obs - 1), T.r[n_obs - 1], new_pos);
      if (active) {
        p = new_pos;
        v = nv;
        PMAF_BOUND(n < D.cap);
      if (sub == 0) { path[n * 3] = p.x; path[n * 3 + 1] = p.y; path[n * 3 + 2] = p.z; }
        n++;
      }
    }
  }
  if (chosen) {
    for (int t = 0; t < ntiles; t++) {
      int i = t * LPA + sub;
      if (i < M) ko[i] = (int32_t)((known_bits >> t) & 1ull);
    }
    if (sub == 0) {
      D.n_points[pa] = n;
      D.agent_vel[pa * 3] = v.x; D.agent_vel[pa * 3 + 1] = v.y; D.agent_vel[pa * 3 + 2] = v.z;
      D.min_obs[pa] = min_obs;
      D.goal_dist[pa] = norm(goal - p);
      if (A.calls_out && A.only) A.calls_out[pop] = calls;
    }
  }
}

// CfManager::setEEAgentPositions (B/src/cf_manager.cpp:220-224) / setEEAgentPosAndVels (:238-244): every agent's
// path restarts at pos (CfAgent::setPosition = clear + push_back); with vel, CfAgent::setVelocity's clamp (:54-61).
// The population-wide rollout start state follows (the next startPrediction rolls out from there).
__global__ void k_set_agents(DevView D, const double *pos, const double *vel) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.P * D.N) return;
  const int pop = idx / D.N;
  const V3 q = mk(pos[pop * 3], pos[pop * 3 + 1], pos[pop * 3 + 2]);
  double *path = D.paths + (size_t)idx * D.cap * 3;
  path[0] = q.x; path[1] = q.y; path[2] = q.z;
  D.n_points[idx] = 1;
  V3 w = mk(0.0, 0.0, 0.0);
  if (vel) {
    w = mk(vel[pop * 3], vel[pop * 3 + 1], vel[pop * 3 + 2]);
    const double vn = norm(w);
    if (vn > D.C.vel_max) w = (D.C.vel_max / vn) * w;
    D.agent_vel[idx * 3] = w.x; D.agent_vel[idx * 3 + 1] = w.y; D.agent_vel[idx * 3 + 2] = w.z;
  }
  if (idx % D.N == 0) {
    D.start_pos[pop * 3] = q.x; D.start_pos[pop * 3 + 1] = q.y; D.start_pos[pop * 3 + 2] = q.z;
    if (vel) { D.start_vel[pop * 3] = w.x; D.start_vel[pop * 3 + 1] = w.y; D.start_vel[pop * 3 + 2] = w.z; }
  }
}

// CfAgent::evalObstacleDistance, B/src/cf_agent.cpp:146-157, one thread per agent: min over ALL obstacles of the
// unclamped surface distance from the agent's latest position, starting from the shell radius
__global__ void k_eval_obstacle_distance(DevView D, const double *obs, double *out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.P * D.N) return;
  const int pop = idx / D.N;
  const int n_obs = D.n_obs;
  const double *o = obs + (size_t)pop * 7 * n_obs;
  const double *path = D.paths + (size_t)idx * D.cap * 3;
  const int n = D.n_points[idx];
  const V3 p = mk(path[(n - 1) * 3], path[(n - 1) * 3 + 1], path[(n - 1) * 3 + 2]);
  double min_dist = D.C.shell;
  for (int k = 0; k < n_obs; k++) {
    const double d = norm(p - mk(o[k], o[n_obs + k], o[2 * n_obs + k])) - (D.C.rad + o[6 * n_obs + k]);
    if (min_dist > d) min_dist = d;
  }
  out[idx] = min_dist;
}

// ---------------------------------------------------------------------------
// launch interface (pmaf_types.hpp)
// ---------------------------------------------------------------------------
bool pmaf_k_launch_w64_m0(const DevView &, const CostParams &, int, bool, bool, size_t, hipStream_t, hipEvent_t, hipEvent_t, bool);
bool pmaf_k_launch_w64_m1(const DevView &, const CostParams &, int, bool, bool, size_t, hipStream_t, hipEvent_t, hipEvent_t, bool);
bool pmaf_k_launch_w64_m2_t1(const DevView &, const CostParams &, int, bool, bool, size_t, hipStream_t, hipEvent_t, hipEvent_t, bool);   // one slot per lane
bool pmaf_k_launch_w64_m2_tn(const DevView &, const CostParams &, int, bool, bool, size_t, hipStream_t, hipEvent_t, hipEvent_t, bool);   // two / four slots
bool pmaf_k_launch_grp_m0(const DevView &, const CostParams &, int, int, int, size_t, hipStream_t, hipEvent_t, hipEvent_t);
bool pmaf_k_launch_grp_m2(const DevView &, const CostParams &, int, int, int, size_t, hipStream_t, hipEvent_t, hipEvent_t);
bool pmaf_k_launch_w64_m3(const DevView &, const CostParams &, int, bool, bool, size_t, hipStream_t, hipEvent_t, hipEvent_t, bool);      // contracted policy
bool pmaf_k_launch_grp_m3(const DevView &, const CostParams &, int, int, int, size_t, hipStream_t, hipEvent_t, hipEvent_t);

bool pmaf_k_launch_w64(const DevView &D, const CostParams &cp, int tiles, int math, bool dppsum, bool plain, size_t lds,
                       hipStream_t s, hipEvent_t e0, hipEvent_t e1, bool slice) {
  if (math == MATH_FMA) return pmaf_k_launch_w64_m3(D, cp, tiles, dppsum, plain, lds, s, e0, e1, slice);
  if (math == MATH_FAST) return pmaf_k_launch_w64_m1(D, cp, tiles, dppsum, plain, lds, s, e0, e1, slice);
  if (math == MATH_IEEE) return pmaf_k_launch_w64_m0(D, cp, tiles, dppsum, plain, lds, s, e0, e1, slice);
  return (tiles <= 1) ? pmaf_k_launch_w64_m2_t1(D, cp, tiles, dppsum, plain, lds, s, e0, e1, slice)
                      : pmaf_k_launch_w64_m2_tn(D, cp, tiles, dppsum, plain, lds, s, e0, e1, slice);
}

bool pmaf_k_launch_mw_m1(const DevView &, const CostParams &, int, int, bool, int, hipStream_t, hipEvent_t, hipEvent_t);
bool pmaf_k_launch_mw_m2(const DevView &, const CostParams &, int, int, bool, int, hipStream_t, hipEvent_t, hipEvent_t);
bool pmaf_k_launch_mw_m3(const DevView &, const CostParams &, int, int, bool, int, hipStream_t, hipEvent_t, hipEvent_t);
bool pmaf_k_launch_mw(const DevView &D, const CostParams &cp, int waves, int per, int math, bool plain, int lds_kb,
                      hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  if (math == MATH_FMA) return pmaf_k_launch_mw_m3(D, cp, waves, per, plain, lds_kb, s, e0, e1);
  if (math == MATH_XACT) return pmaf_k_launch_mw_m2(D, cp, waves, per, plain, lds_kb, s, e0, e1);
  if (math == MATH_FAST) return pmaf_k_launch_mw_m1(D, cp, waves, per, plain, lds_kb, s, e0, e1);
  return false;
}

bool pmaf_k_launch_grp(const DevView &D, const CostParams &cp, int lpa, int tiles, int math, int n_blocks, size_t lds,
                       hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  if (math == MATH_IEEE) return pmaf_k_launch_grp_m0(D, cp, lpa, tiles, n_blocks, lds, s, e0, e1);
  if (math == MATH_FMA) return pmaf_k_launch_grp_m3(D, cp, lpa, tiles, n_blocks, lds, s, e0, e1);
  return pmaf_k_launch_grp_m2(D, cp, lpa, tiles, n_blocks, lds, s, e0, e1);
}

bool pmaf_k_launch_generic(const DevView &D, const CostParams &cp, int lpa, int n_blocks, size_t lds, hipStream_t s,
                           hipEvent_t e0, hipEvent_t e1) {
  const dim3 grid((unsigned)n_blocks, (unsigned)D.P), block(64);
  switch (lpa) {
#define PMAF_CASE(L) case L: hipExtLaunchKernelGGL((k_rollout<L>), grid, block, (unsigned)lds, s, e0, e1, 0, D, cp); break;
    PMAF_CASE(1) PMAF_CASE(2) PMAF_CASE(4) PMAF_CASE(8) PMAF_CASE(16) PMAF_CASE(32) PMAF_CASE(64)
#undef PMAF_CASE
    default: return false;
  }
  return true;
}

void pmaf_k_launch_manager(const DevView &D, const CostParams &cp, const ManagerArgs &A, size_t lds, hipStream_t s,
                           hipEvent_t done) {
  hipExtLaunchKernelGGL(k_manager, dim3((unsigned)D.P), dim3(64), (unsigned)lds, s, nullptr, done, 0, D, cp, A);
}

void pmaf_k_launch_score(const DevView &D, const CostParams &cp, hipStream_t s) {
  const int total = D.P * D.N;
  hipLaunchKernelGGL(k_score, dim3((total + 63) / 64), dim3(64), 0, s, D, cp);
}

void pmaf_k_launch_restart_paths(const DevView &D, const double *pos, hipStream_t s) {
  hipLaunchKernelGGL(k_restart_paths, dim3((D.P * D.N + 255) / 256), dim3(256), 0, s, D, pos);
}

void pmaf_k_launch_link_force(int n, const double *link_pos, const double *k_r, const double *sent, double rad,
                              double shell, double *out, hipStream_t s) {
  hipLaunchKernelGGL(k_link_force, dim3((n + 63) / 64), dim3(64), 0, s, n, link_pos, k_r, sent, rad, shell, out);
}

void pmaf_k_launch_debug_math(int op, int n, const double *a, const double *b, double *out, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, s, op, n, a, b, out);
}

void pmaf_k_launch_winner(const DevView &D, double *dst, hipStream_t s) {
  hipLaunchKernelGGL(k_winner, dim3((unsigned)D.P), dim3(256), 0, s, D, dst);
}

void pmaf_k_launch_winner_path(const DevView &D, const double *paths, double *dst, hipStream_t s) {
  hipLaunchKernelGGL(k_winner_path, dim3((unsigned)D.P), dim3(256), 0, s, D, paths, dst);
}

bool pmaf_k_launch_plan_steps(const DevView &D, const PlanArgs &A, int lpa, int n_blocks, size_t lds, hipStream_t s) {
  const dim3 grid((unsigned)n_blocks, (unsigned)D.P), block(64);
  switch (lpa) {
#define PMAF_CASE(L) case L: hipLaunchKernelGGL((k_plan_steps<L>), grid, block, lds, s, D, A); break;
    PMAF_CASE(1) PMAF_CASE(2) PMAF_CASE(4) PMAF_CASE(8) PMAF_CASE(16) PMAF_CASE(32) PMAF_CASE(64)
#undef PMAF_CASE
    default: return false;
  }
  return true;
}

void pmaf_k_launch_set_agents(const DevView &D, const double *pos, const double *vel, hipStream_t s) {
  hipLaunchKernelGGL(k_set_agents, dim3((D.P * D.N + 255) / 256), dim3(256), 0, s, D, pos, vel);
}

void pmaf_k_launch_eval_obstacle_distance(const DevView &D, const double *obs, double *out, hipStream_t s) {
  hipLaunchKernelGGL(k_eval_obstacle_distance, dim3((D.P * D.N + 255) / 256), dim3(256), 0, s, D, obs, out);
}

hipError_t pmaf_k_set_lds_limits(size_t lds_manager, size_t lds_rollout) {
  hipError_t e = hipSuccess;
  // the attribute belongs to the kernel, not to a handle: a second handle with a smaller (still > 64 KB) table must not
  // lower the limit under the first one's launches -- the largest request of the process stands (per device: the
  // attribute is set on the current device's copy of the function; a handle re-raises it on its own device)
  static size_t max_manager[16] = {0}, max_rollout[16] = {0};
  static std::mutex mtx;                      // (handles may be created from different threads)
  std::lock_guard<std::mutex> lock(mtx);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16) {
    if (lds_manager < max_manager[dev]) lds_manager = max_manager[dev]; else max_manager[dev] = lds_manager;
    if (lds_rollout < max_rollout[dev]) lds_rollout = max_rollout[dev]; else max_rollout[dev] = lds_rollout;
  }
  if (lds_manager > 64 * 1024)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_manager), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_manager);
  if (e == hipSuccess && lds_rollout > 64 * 1024) {
    // only the generic kernel takes obstacle tables this large (M > 256)
#define PMAF_LDS(L) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rollout); \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_plan_steps<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rollout);
    PMAF_LDS(1) PMAF_LDS(2) PMAF_LDS(4) PMAF_LDS(8) PMAF_LDS(16) PMAF_LDS(32) PMAF_LDS(64)
#undef PMAF_LDS
  }
  return e;
}

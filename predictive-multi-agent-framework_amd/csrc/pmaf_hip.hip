// pmaf_hip.hip -- libpmaf_hip.so: HIP kernels (gfx950) + C-ABI (include/pmaf.h)
// of the predictive multi-agent circular-field planner tick.
//
// Kernels
//   k_rollout<LPA>  agent x horizon rollout (CfAgent::cfPrediction,
//                   B/src/cf_agent.cpp:302-341) for all agents of all
//                   populations in one launch, with the per-path cost terms of
//                   CfManager::evaluateAgents (B/src/cf_manager.cpp:302-333)
//                   accumulated on the fly. One wave64 per block, 64/LPA agents
//                   per wave, obstacle table in LDS advanced once per step.
//   k_manager       one wave per population: evaluateAgents' cost assembly +
//                   argmin + hysteresis (:325-353), RealCfAgent::cfPlanner
//                   single step (B/src/cf_agent.cpp:343-366) and
//                   resetEEAgents (B/src/cf_manager.cpp:246-255).
//   k_score         re-scores stored paths (used before the first rollout and
//                   when evaluate is called with other workspace gains).
//   k_link_force    CfAgent::bodyForce (B/src/cf_agent.cpp:229-234).
//   k_winner        packs winner records for sharded runs.
// There is no CPU fallback in this library.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pmaf.h"
#include "pmaf_device.hpp"
#include "pmaf_rollout_w64.hpp"
#include "pmaf_rollout_grp.hpp"

using namespace pmaf;

// ---------------------------------------------------------------------------
// device-side views
// ---------------------------------------------------------------------------
struct DevView {
  int P, N, n_obs, cap;
  PopConst C;
  // per population
  const double *goal;        // [P][3]
  double *agent_init_pos;    // [P][3]  CfAgent::init_pos_ (gate)
  double *start_pos;         // [P][3]  position all agents start the next rollout from
  double *start_vel;         // [P][3]
  double *obs_start;         // [P][7][n_obs] SoA: agents' private obstacle copies at rollout start
  int32_t *known_start;      // [P][n_obs]
  double *obs_live;          // [P][7][n_obs] SoA: live obstacles (moveRealEEAgent / resetEEAgents argument)
  // per agent
  const double *k_attr, *k_circ, *k_repel, *k_damp;  // [P][N]
  const int32_t *types;      // [N]
  double *rot;               // [P][N][3][n_obs]  field_rotation_vecs_
  const double *rnd;         // [P][N][3][n_obs]  random_vecs_
  double *paths;             // [P][N][cap][3]
  int32_t *n_points;         // [P][N]
  double *agent_vel;         // [P][N][3]
  double *min_obs;           // [P][N]
  double *cost_ws;           // [P][N]  sum of workspace penalties over the path
  double *path_len;          // [P][N]
  double *goal_dist;         // [P][N]
  int32_t *reached;          // [P][N]
  int32_t *known_out;        // [P][N][n_obs] known_obstacles_ after the rollout
  double *costs;             // [P][N]
  // real agent
  double *real_pos, *real_vel, *real_force, *real_init_pos;  // [P][3]
  int32_t *real_known;       // [P][n_obs]
  double *real_rot;          // [P][3][n_obs]
  // best agent copy
  int32_t *has_best, *best_id, *best_type;  // [P]
  double *best_rnd;          // [P][3][n_obs]
  int32_t *best_idx;         // [P] last evaluate result
  unsigned long long *step_counter;  // [1] agent-steps executed by all rollouts
  unsigned long long *pred_ticks;    // [P][N] rollout duration in wall_clock64() ticks (CfAgent::prediction_time_)
  const double *zsent_lt;            // [P] exact squared-distance boundary of the repel range test
  int ablate;                        // timing experiments only (PMAF_ABLATE), 0 in production
  int n_simds;                       // SIMDs of the device (4 per CU): a launch of more waves doubles them up
};

struct CostParams {
  double k_goal_dist, k_path_len, k_safe_dist, k_workspace;
  double ws[6];
};

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
// k_rollout_w64<TILES>: one wave64 per agent (see pmaf_rollout_w64.hpp)
// ---------------------------------------------------------------------------
// the step loop, specialised on the agent's heuristic so the per-step code
// carries no type dispatch (the type is uniform per wave)
// SENT: 0 = the repulsive obstacle cannot come into range during this rollout (decided by the caller, one-slot kernel
// only: the step loop then has no block for it), 1 = it can, 2 = decide here at run time (the other kernels)
template <int TILES, int TYPE, int MATH, int SENT = 2>
__device__ __forceinline__ void rollout_w64_body(const DevView &D, const CostParams &CP, const int lane,
                                                 const int pop, const int a) {
  extern __shared__ double smem[];
  const unsigned long long t_begin = wall_clock64();
  const int n_obs = D.n_obs;
  const int M = n_obs - 1;
  // Loop-invariant wave-uniform doubles (population constants, goal, gains, the exp() coefficients) are pinned
  // into VGPRs: left to itself the compiler keeps them in the 100-odd SGPRs, runs out, and pays for the
  // spills (v_writelane / v_readlane) in the step loop -- 163 spilled SGPRs before, C2 360 -> 344 us with this.
  PopConst C = D.C;
  { double *f = reinterpret_cast<double *>(&C);
    for (int i = 0; i < (int)(sizeof(PopConst) / sizeof(double)); i++) asm volatile("" : "+v"(f[i])); }
  // (two and four slots per lane have no VGPRs to spare for the exp coefficients: measured)
  const ExpK EK = (TILES >= 2) ? exp_consts() : exp_consts_in_vgprs();
  const size_t pa = (size_t)pop * D.N + a;
  const double *src = D.obs_start + (size_t)pop * 7 * n_obs;
  const int32_t *ks = D.known_start + (size_t)pop * n_obs;
  double *rot_g = D.rot + pa * 3 * n_obs;
  const double *rnd_g = D.rnd + pa * 3 * n_obs;

  LaneObstacles<TILES> O;
  unsigned known_bits = 0u;
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    int i = t * 64 + lane;
    bool valid = i < M;
    int ii = valid ? i : 0;
    O.p[t] = mk(src[ii], src[n_obs + ii], src[2 * n_obs + ii]);
    O.v[t] = mk(src[3 * n_obs + ii], src[4 * n_obs + ii], src[5 * n_obs + ii]);
    O.r[t] = src[6 * n_obs + ii];
    O.rx[t] = rot_g[ii]; O.ry[t] = rot_g[n_obs + ii]; O.rz[t] = rot_g[2 * n_obs + ii];
    if (TYPE == T_RANDOM) { O.qx[t] = rnd_g[ii]; O.qy[t] = rnd_g[n_obs + ii]; O.qz[t] = rnd_g[2 * n_obs + ii]; }
    else { O.qx[t] = 0.0; O.qy[t] = 0.0; O.qz[t] = 0.0; }
    if (valid && ks[ii]) known_bits |= (1u << t);
  }
  // trailing repulsive obstacle, wave-uniform
  V3 sent_p = mk(src[M], src[n_obs + M], src[2 * n_obs + M]);
  const V3 sent_v = mk(src[3 * n_obs + M], src[4 * n_obs + M], src[5 * n_obs + M]);
  const double sent_r = src[6 * n_obs + M];

  V3 goal = mk(D.goal[pop * 3], D.goal[pop * 3 + 1], D.goal[pop * 3 + 2]);
  V3 init_pos = mk(D.agent_init_pos[pop * 3], D.agent_init_pos[pop * 3 + 1], D.agent_init_pos[pop * 3 + 2]);
  asm volatile("" : "+v"(goal.x), "+v"(goal.y), "+v"(goal.z), "+v"(init_pos.x), "+v"(init_pos.y), "+v"(init_pos.z));
  V3 p = mk(D.start_pos[pop * 3], D.start_pos[pop * 3 + 1], D.start_pos[pop * 3 + 2]);
  V3 v = mk(D.start_vel[pop * 3], D.start_vel[pop * 3 + 1], D.start_vel[pop * 3 + 2]);
  double k_attr = D.k_attr[pa], k_circ = D.k_circ[pa], k_repel = D.k_repel[pa], k_damp = D.k_damp[pa];
  asm volatile("" : "+v"(k_attr), "+v"(k_circ), "+v"(k_repel), "+v"(k_damp));
  double *path = D.paths + pa * (size_t)D.cap * 3;

  // LDS list of the step's non-zero circular-field terms (after the obstacle table)
  int clist_off = 7 * n_obs + (n_obs + 1) / 2;
  clist_off += clist_off & 1;
  double *clist = smem + clist_off;

  double lane_min = C.shell;  // per-lane running min_obs_dist_, reduced once after the loop
  int n = 1;
  bool ran = false;
  if (lane == 0) { path[0] = p.x; path[1] = p.y; path[2] = p.z; }

  // Everything the next step needs from the new state -- goal distance (loop
  // guard / gate), goal direction, squared speed, squared start distance and
  // attractorForce's velocity error (B/src/cf_agent.cpp:188-192) -- is computed
  // at the END of the step in one basic block with the velocity clamp and the
  // path-length norm: five independent sqrt / divide chains that the in-order
  // issue of a lone wave can only overlap inside one block.
  typedef Mth<MATH> MT;
  V3 g = goal - p;
  double dg = MT::norm(g);
  double zv = sqn(v);
  double z_init = sqn(p - init_pos);
  V3 gn = (dg > 0.0) ? MT::div3(g, dg) : g;  // goal_vec.normalized()
  // One slot per lane (M <= 61; the host sends 62..64 obstacles to the two-slot kernel): the sweep's |ro| /
  // ro.normalized() of the NEXT step are computed at the end of this step, and the lanes that have no obstacle carry
  // the tail's other norms through the same instructions -- lane 63 the goal (distance and direction), lane 62 the
  // speed clamp, lane 61 attractorForce's speed limit -- instead of three more sqrt / reciprocal / divide sequences.
  constexpr bool PRE = (TILES == 1);
  double s_pre = 0.0;
  V3 ron_pre = mk(0.0, 0.0, 0.0);
  if (PRE) {
    if (lane == 63) { O.p[0] = goal; O.v[0] = mk(0.0, 0.0, 0.0); }
    MT::norm_unit(O.p[0] - p, s_pre, ron_pre);
  }
  V3 verr = attractor_velocity_error<MATH>(v, g, C, k_attr, k_damp);
  const double zsent_lt = D.zsent_lt[pop];
  const bool sent_reachable = (SENT == 2) ? sentinel_reachable(p, sent_p, sent_v, zsent_lt, C, D.cap) : (SENT == 1);
  bool moving = false;  // any field obstacle with a non-zero (or NaN) velocity component
#pragma unroll
  for (int t = 0; t < TILES; t++) moving = moving || !(O.v[t].x == 0.0 && O.v[t].y == 0.0 && O.v[t].z == 0.0);
  moving = __any(moving);
  bool advance = true;
  V3 repel = mk(0.0, 0.0, 0.0);  // repelForce of the coming step (depends on the step's start state only)
  if (sent_reachable) repel = sentinel_repel(p, C, k_repel, sent_p, sent_r, zsent_lt);
  SecTimers ST;
#ifdef PMAF_SECTION_TIMERS
  ST.start();
#endif
  while ((dg > 0.1) && (n < D.cap)) {  // wave-uniform guard, B/src/cf_agent.cpp:310-311
    // gate, :315-317
    // |v| < 0.5 vmax and |p - init| < 0.2 on exact squared thresholds
    const bool gate = !(dg < C.approach || (zv < C.zvhalf_lt && z_init < C.zinit_lt));
    V3 F = mk(0.0, 0.0, 0.0);
    double scale = 1.0;
    PMAF_SEC(ST, 0);
    // (called with the gate closed too: the sweep's few compares then find no obstacle -- one branch less in the step)
    if (PRE || (gate && !(D.ablate & 8)))
      circ_and_scale_w64<TILES, TYPE, MATH, PRE>(lane, p, v, zv, goal, g, dg, gn, C, k_circ, n_obs, rot_g, known_bits,
                                                 O, clist, lane_min, F, scale, ST, EK, D.ablate, 0, s_pre, ron_pre,
                                                 gate);
    PMAF_SEC(ST, 5);
    // attractorForce (:183-193), updatePositionAndVelocity (:253-268)
    // repelForce (:159-181): `repel` was evaluated for this step's start state at the end of the previous step; it is
    // +0.0 when the obstacle cannot come into range, and F -- a sum that started from +0.0 -- is never -0.0, so the
    // unconditional addition is exact (no masked block between the force sum and the tail)
    F = F + (mk(0.0, 0.0, 0.0) + repel);
    // attractorForce (:183-193), updatePositionAndVelocity (:253-258): a = F / mass, |a| <= 13
    V3 acc;
    if (PRE) {
      // one slot per lane: as few blocks as possible between the force sum and the tail -- the k_attr == 0 case is a
      // select, and unit mass without clamp (the common case) skips ONE rare block instead of two
      const V3 Fa = F + (scale * k_damp) * verr;
      const bool attr = (k_attr != 0.0);
      F.x = attr ? Fa.x : F.x; F.y = attr ? Fa.y : F.y; F.z = attr ? Fa.z : F.z;
      acc = F;
      double az = sqn(F);
      if ((C.mass != 1.0) || (az >= C.zacc_gt)) {
        if (C.mass != 1.0) { acc = F / C.mass; az = sqn(acc); }
        if (az >= C.zacc_gt) acc = acc * (13.0 / __builtin_sqrt(az));  // norm(acc) > 13.0
      }
    } else {
      if (k_attr != 0.0) F = F + (scale * k_damp) * verr;
      acc = F;
      if (C.mass != 1.0) acc = F / C.mass;
      const double az = sqn(acc);
      if (az >= C.zacc_gt) acc = acc * (13.0 / __builtin_sqrt(az));  // norm(acc) > 13.0 (rare)
    }
    PMAF_SEC(ST, 6);
    // ---- one block: integrate, clamp the speed, the next step's norms ----
    const V3 half = ((0.5 * acc) * C.dt) * C.dt;
    const V3 new_pos = (p + half) + (v * C.dt);
    const V3 nv = v + acc * C.dt;
    p = new_pos;
    g = goal - p;
    // predictObstacles, B/src/cf_agent.cpp:270-276, in registers. Obstacles at
    // rest: p + (+-0) dt is idempotent after its first application (which turns
    // a -0.0 coordinate into +0.0), so later steps skip it. (Lane 63 of the one-slot kernel holds the goal with
    // velocity 0: a -0.0 goal coordinate turns into +0.0 there, which can only change the sign of a zero component
    // of gn, and gn only enters dot(ron, gn) < -0.01; g itself is computed from the goal directly.)
    // (one slot per lane: unconditionally -- for obstacles at rest every further application is the identity, and
    // three multiply-adds are cheaper than a branch in the middle of the tail)
    if (PRE) O.p[0] = O.p[0] + O.v[0] * C.dt;
    {
      // ONE sqrt / reciprocal / divide sequence for the whole tail: lane 63 goal distance and direction (|g|,
      // g.normalized()), lane 62 the speed clamp (|nv|, vel_max / |nv|), lane 61 attractorForce's limit
      // (vel_max / |vel_des|) and, in the one-slot kernel, lanes 0..M-1 the next step's |ro| and ro.normalized():
      // the same operations on the same operands as separate sequences, read back with v_readlane.
      const V3 vel_des = (k_attr / k_damp) * g;
      const bool l_nv = (lane == 62), l_des = (lane == 61);
      const V3 other = PRE ? (O.p[0] - p) : g;   // one-slot kernel: lane 63 holds the goal, O.p - p = g there
      const V3 vec = l_nv ? nv : (l_des ? vel_des : other);
      V3 num = vec;
      num.x = (l_nv || l_des) ? C.vel_max : vec.x;
      double s, rs;
      MT::norm_rcp(vec, s, rs);
      const V3 q = MT::div3_n(num, s, rs);
      const V3 u = (sqn(vec) > 0.0) ? q : vec;  // normalized(): the vector itself unless squaredNorm > 0
      if (PRE) { s_pre = s; ron_pre = u; }
      const double vn = readlane_d(s, 62), f_nv = readlane_d(q.x, 62), f_des = readlane_d(q.x, 61);
      v = (vn > C.vel_max) ? nv * f_nv : nv;
      dg = readlane_d(s, 63);
      gn = readlane_v3(u, 63);
      verr = vel_des * smin(1.0, f_des) - v;
    }
    zv = sqn(v);
    z_init = sqn(p - init_pos);
    // every lane stores the (wave-uniform) point to the same address: one transaction, and no exec-masked block
    // in the middle of the tail
    path[n * 3] = p.x; path[n * 3 + 1] = p.y; path[n * 3 + 2] = p.z;
    n++;
    ran = true;
    if (!PRE && advance) {
#pragma unroll
      for (int t = 0; t < TILES; t++) O.p[t] = O.p[t] + O.v[t] * C.dt;
      advance = moving;
    }
    if (sent_reachable) {  // the only masked block of the step for the repulsive obstacle: advance it, next step's repelForce
      sent_p = sent_p + sent_v * C.dt;
      repel = sentinel_repel(p, C, k_repel, sent_p, sent_r, zsent_lt);
    }
    PMAF_SEC(ST, 7);
  }
#ifdef PMAF_SECTION_TIMERS
  if (lane == 0 && pop == 0 && a < 7)
    printf("agent %d type %d steps %d | verr+gate %llu sweep %llu scale %llu circ %llu sum %llu (skip) %llu finish %llu tail %llu | "
           "in-shell steps %llu terms %llu\n", a, TYPE, n - 1, ST.acc[0], ST.acc[1], ST.acc[2], ST.acc[3], ST.acc[4],
           ST.acc[5], ST.acc[6], ST.acc[7], ST.cnt[0], ST.cnt[1]);
#endif

  double cost_ws, path_len;
  path_cost_terms_w64<MATH>(lane, path, n, CP.ws, CP.k_workspace, clist, cost_ws, path_len);

  const double min_obs = wave_min64(lane_min);
  int32_t *ko = D.known_out + pa * n_obs;
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    int i = t * 64 + lane;
    if (i < M) ko[i] = (int32_t)((known_bits >> t) & 1u);
  }
  if (lane == 0) {
    ko[M] = ks[M];
    D.n_points[pa] = n;
    D.agent_vel[pa * 3] = v.x; D.agent_vel[pa * 3 + 1] = v.y; D.agent_vel[pa * 3 + 2] = v.z;
    D.min_obs[pa] = min_obs;
    D.cost_ws[pa] = cost_ws;
    D.path_len[pa] = path_len;
    D.goal_dist[pa] = dg;
    if (ran) D.reached[pa] = dg < 0.100001;  // B/src/cf_agent.cpp:330-337
    atomicAdd(D.step_counter, (unsigned long long)(n - 1));
    D.pred_ticks[pa] = wall_clock64() - t_begin;
  }
}

template <int TILES, int MATH>
__global__ __launch_bounds__(64) void k_rollout_w64(DevView D, CostParams CP) {
  const int lane = threadIdx.x;
  const int pop = blockIdx.y;
  const int a = blockIdx.x;  // grid.x == N
  if (TILES == 1) {
    // one slot per lane: the repulsive obstacle's reachability (per rollout, see sentinel_reachable) picks a loop
    // without any code for it -- in the shipped scenes it sits 170 m away
    const int n_obs = D.n_obs, M = n_obs - 1;
    const double *src = D.obs_start + (size_t)pop * 7 * n_obs;
    const V3 sp = mk(src[M], src[n_obs + M], src[2 * n_obs + M]);
    const V3 sv = mk(src[3 * n_obs + M], src[4 * n_obs + M], src[5 * n_obs + M]);
    const V3 p0 = mk(D.start_pos[pop * 3], D.start_pos[pop * 3 + 1], D.start_pos[pop * 3 + 2]);
    const PopConst C0 = D.C;
    const bool reach = sentinel_reachable(p0, sp, sv, D.zsent_lt[pop], C0, D.cap);
#define PMAF_BODY(T) \
    if (reach) rollout_w64_body<TILES, T, MATH, 1>(D, CP, lane, pop, a); \
    else rollout_w64_body<TILES, T, MATH, 0>(D, CP, lane, pop, a)
    switch (D.types[a]) {
      case T_GOAL: PMAF_BODY(T_GOAL); break;
      case T_OBST: PMAF_BODY(T_OBST); break;
      case T_GOALOBST: PMAF_BODY(T_GOALOBST); break;
      case T_VEL: PMAF_BODY(T_VEL); break;
      case T_RANDOM: PMAF_BODY(T_RANDOM); break;
      case T_HAD: PMAF_BODY(T_HAD); break;
      default: break;
    }
#undef PMAF_BODY
    return;
  }
  switch (D.types[a]) {
    case T_GOAL: rollout_w64_body<TILES, T_GOAL, MATH>(D, CP, lane, pop, a); break;
    case T_OBST: rollout_w64_body<TILES, T_OBST, MATH>(D, CP, lane, pop, a); break;
    case T_GOALOBST: rollout_w64_body<TILES, T_GOALOBST, MATH>(D, CP, lane, pop, a); break;
    case T_VEL: rollout_w64_body<TILES, T_VEL, MATH>(D, CP, lane, pop, a); break;
    case T_RANDOM: rollout_w64_body<TILES, T_RANDOM, MATH>(D, CP, lane, pop, a); break;
    case T_HAD: rollout_w64_body<TILES, T_HAD, MATH>(D, CP, lane, pop, a); break;
    default: break;
  }
}

// ---------------------------------------------------------------------------
// k_rollout_grp<LPA, TILES>: 64/LPA agents per wave (see pmaf_rollout_grp.hpp)
// ---------------------------------------------------------------------------
// SIMD sharing in the group kernel (launches of more waves than the device has SIMDs), measured on C5 = 2048 waves:
// * block b and block b + n_simds land on the same SIMD (tools/placement.hip), and the issue arbiter serves the older
//   wave first: the first 1024 waves ran at their stand-alone speed (600 us), the others on the issue slots left over
//   and then alone (880-1000 us; kernel 1010 us). The waves therefore trade issue priority in 10 us slices of the
//   wall clock, the younger wave holding it 5 slices of 8 (the share at which both finish together: 845 us each,
//   kernel 910 us; 4 of 8: 720 / 860, 6 of 8: 860 / 750).
// * every population's first two waves hold its five heuristic agents (mixed types: the wave runs the union of their
//   code paths, 1.13x the work of a wave of Random agents), and b + n_simds is the same wave index of another
//   population: the wave index is rotated by 8 per population so that two such waves do not share a SIMD (-3 %).
constexpr int PRIO_SLICE_LOG2 = 10;       // 2^10 ticks of the 100 MHz wall clock
constexpr unsigned PRIO_YOUNGER_OF_8 = 5;
constexpr unsigned POP_ROTATE = 8;
template <int LPA, int TILES, int MATH>
__global__ __launch_bounds__(64) void k_rollout_grp(DevView D, CostParams CP) {
  extern __shared__ double smem[];
  constexpr int APW = 64 / LPA;
  const unsigned long long t_begin = wall_clock64();
  const int lane = threadIdx.x;
  const int pop = blockIdx.y;
  const int sub = lane % LPA;
  const int grp = lane / LPA;
  const int bx = (int)((blockIdx.x + gridDim.x - (POP_ROTATE * blockIdx.y) % gridDim.x) % gridDim.x);  // see above
  const int a = bx * APW + grp;
  const bool active = a < D.N;
  const int aa = active ? a : 0;
  const int n_obs = D.n_obs;
  const int M = n_obs - 1;
  const PopConst C = D.C;
  const size_t pa = (size_t)pop * D.N + aa;
  const int type = D.types[aa];
  const double *src = D.obs_start + (size_t)pop * 7 * n_obs;
  const int32_t *ks = D.known_start + (size_t)pop * n_obs;
  double *rot_g = D.rot + pa * 3 * n_obs;
  const double *rnd_g = D.rnd + pa * 3 * n_obs;

  LaneObstacles<TILES> O;
  unsigned known_bits = 0u;
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    int i = t * LPA + sub;
    bool valid = i < M;
    int ii = valid ? i : 0;
    O.p[t] = mk(src[ii], src[n_obs + ii], src[2 * n_obs + ii]);
    O.v[t] = mk(src[3 * n_obs + ii], src[4 * n_obs + ii], src[5 * n_obs + ii]);
    O.r[t] = src[6 * n_obs + ii];
    O.rx[t] = rot_g[ii]; O.ry[t] = rot_g[n_obs + ii]; O.rz[t] = rot_g[2 * n_obs + ii];
    O.qx[t] = rnd_g[ii]; O.qy[t] = rnd_g[n_obs + ii]; O.qz[t] = rnd_g[2 * n_obs + ii];
    if (valid && ks[ii]) known_bits |= (1u << t);
  }
  V3 sent_p = mk(src[M], src[n_obs + M], src[2 * n_obs + M]);
  const V3 sent_v = mk(src[3 * n_obs + M], src[4 * n_obs + M], src[5 * n_obs + M]);
  const double sent_r = src[6 * n_obs + M];
  wave_lds_fence();

  const V3 goal = mk(D.goal[pop * 3], D.goal[pop * 3 + 1], D.goal[pop * 3 + 2]);
  const V3 init_pos = mk(D.agent_init_pos[pop * 3], D.agent_init_pos[pop * 3 + 1], D.agent_init_pos[pop * 3 + 2]);
  V3 p = mk(D.start_pos[pop * 3], D.start_pos[pop * 3 + 1], D.start_pos[pop * 3 + 2]);
  V3 v = mk(D.start_vel[pop * 3], D.start_vel[pop * 3 + 1], D.start_vel[pop * 3 + 2]);
  const double k_attr = D.k_attr[pa], k_circ = D.k_circ[pa], k_repel = D.k_repel[pa], k_damp = D.k_damp[pa];
  double *path = D.paths + pa * (size_t)D.cap * 3;
  const double zsent_lt = D.zsent_lt[pop];
  const bool sent_reachable = __any(sentinel_reachable(p, sent_p, sent_v, zsent_lt, C, D.cap));  // wave-uniform
  bool moving = false;
#pragma unroll
  for (int t = 0; t < TILES; t++) moving = moving || !(O.v[t].x == 0.0 && O.v[t].y == 0.0 && O.v[t].z == 0.0);
  moving = __any(moving);
  bool advance = true;

  int clist_off = 7 * n_obs + (n_obs + 1) / 2;
  clist_off += clist_off & 1;
  double *clist = smem + clist_off + (size_t)grp * ((LPA * TILES + 1) * 4);
  if (sub < 4) clist[(size_t)LPA * TILES * 4 + sub] = 0.0;  // the group's all-zero list entry
  wave_lds_fence();

  double lane_min = C.shell;
  int n = 1;
  bool ran = false;
  if (active && sub == 0) { path[0] = p.x; path[1] = p.y; path[2] = p.z; }

  V3 g = goal - p;
  double dg = Mth<MATH>::norm(g);
  double zv = sqn(v);
  double z_init = sqn(p - init_pos);
  const unsigned lin_block = blockIdx.y * gridDim.x + blockIdx.x;
  const bool shared_simd = gridDim.x * gridDim.y > (unsigned)D.n_simds;
  const bool younger = ((lin_block / (unsigned)D.n_simds) & 1u) != 0u;
  while (true) {
    const bool run = active && (dg > 0.1) && (n < D.cap);  // B/src/cf_agent.cpp:310-311, per agent
    if (!__any(run)) break;
    unsigned long long clk = 0ull;
    if (shared_simd) clk = wall_clock64();
    const bool gate = !(dg < C.approach || (zv < C.zvhalf_lt && z_init < C.zinit_lt));  // :315-317
    const V3 verr = attractor_velocity_error<MATH>(v, g, C, k_attr, k_damp);
    V3 F = mk(0.0, 0.0, 0.0);
    double scale = 1.0;
    if (__any(run && gate))
      circ_and_scale_grp<LPA, TILES, MATH>(run && gate, sub, grp, type, p, v, zv, goal, g, dg, C, k_circ, n_obs, rot_g,
                                     known_bits, O, clist, lane_min, F, scale);
    V3 new_pos;
    V3 nv = v;
    finish_step_w64<MATH>(p, nv, verr, F, scale, C, k_attr, k_repel, k_damp, sent_p, sent_r, zsent_lt, new_pos,
                          sent_reachable);
    if (run) {
      p = new_pos;
      v = nv;
      g = goal - p;
      dg = Mth<MATH>::norm(g);
      zv = sqn(v);
      z_init = sqn(p - init_pos);
      if (sub == 0) { path[n * 3] = p.x; path[n * 3 + 1] = p.y; path[n * 3 + 2] = p.z; }
      n++;
      ran = true;
    }
    // predictObstacles, B/src/cf_agent.cpp:270-276, in registers (obstacles at rest: once, see the w64 kernel)
    if (advance) {
#pragma unroll
      for (int t = 0; t < TILES; t++) O.p[t] = O.p[t] + O.v[t] * C.dt;
      advance = moving;
    }
    if (sent_reachable) sent_p = sent_p + sent_v * C.dt;
    if (shared_simd) {
      if (((((unsigned)(clk >> PRIO_SLICE_LOG2)) & 7u) < PRIO_YOUNGER_OF_8) == younger) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
  }
  __builtin_amdgcn_s_setprio(0);

  double cost_ws, path_len;
  path_cost_terms_grp<LPA, MATH>(sub, grp, active, path, n, CP.ws, CP.k_workspace, clist, cost_ws, path_len);

  const double min_obs = group_min_dpp<LPA>(lane_min);
  if (active) {
    int32_t *ko = D.known_out + pa * n_obs;
#pragma unroll
    for (int t = 0; t < TILES; t++) {
      int i = t * LPA + sub;
      if (i < M) ko[i] = (int32_t)((known_bits >> t) & 1u);
    }
    if (sub == 0) {
      ko[M] = ks[M];
      D.n_points[pa] = n;
      D.agent_vel[pa * 3] = v.x; D.agent_vel[pa * 3 + 1] = v.y; D.agent_vel[pa * 3 + 2] = v.z;
      D.min_obs[pa] = min_obs;
      D.cost_ws[pa] = cost_ws;
      D.path_len[pa] = path_len;
      D.goal_dist[pa] = dg;
      if (ran) D.reached[pa] = dg < 0.100001;  // B/src/cf_agent.cpp:330-337
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
                                              unsigned &known_bits) {
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
  const V3 repel = sentinel_repel(rp, C, k_repel, sent_p, sent_r, D.zsent_lt[pop]);
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

struct ManagerArgs {
  int do_select, do_move, do_reset;
  int reset_from_real;     // 1: reset to the real agent's state, 0: reset_in
  int rollout_follows;     // 1: the rollout kernel is launched right behind this one (pmaf_tick)
  int tuned_real_step;     // 1: real_step_w64 (default arithmetic policy, M <= 256), 0: generic LDS-table path
  const double *live_src;  // [P][7][n_obs] live obstacles in mapped pinned HOST memory (pmaf_tick: no copy command);
                           // NULL: D.obs_live already holds them
  double dt_real;
  const int32_t *agent_id; // [P] gains index for the real step; NULL = best_idx of this launch
  const double *reset_in;  // [P][6] pos, vel
  double *out;             // [P][12] host-visible: best_idx, next_pos[3], next_vel[3], dist_from_goal, force[3], seq
  double seq;              // written to out[11] after the other entries are visible to the host (0: not written)
};

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
  const V3 goal = mk(D.goal[pop * 3], D.goal[pop * 3 + 1], D.goal[pop * 3 + 2]);
  int best = D.best_idx[pop];
  const int had_best = D.has_best[pop];
  const int old_best_id = D.best_id[pop];
  int htype = D.best_type[pop];
  V3 rp = mk(D.real_pos[pop * 3], D.real_pos[pop * 3 + 1], D.real_pos[pop * 3 + 2]);
  V3 rv = mk(D.real_vel[pop * 3], D.real_vel[pop * 3 + 1], D.real_vel[pop * 3 + 2]);
  V3 rf = mk(D.real_force[pop * 3], D.real_force[pop * 3 + 1], D.real_force[pop * 3 + 2]);
  const V3 init_pos = mk(D.real_init_pos[pop * 3], D.real_init_pos[pop * 3 + 1], D.real_init_pos[pop * 3 + 2]);
  int32_t *rk = D.real_known + (size_t)pop * n_obs;
  // live obstacles: read straight from the caller's pinned staging buffer when the tick brought new ones (1.8 KB
  // over PCIe costs less than a copy command in front of this kernel) and kept in D.obs_live for later calls
  double *live_dev = D.obs_live + (size_t)pop * 7 * n_obs;
  const double *live = A.live_src ? A.live_src + (size_t)pop * 7 * n_obs : live_dev;
  if (A.do_move || A.do_reset) {
    for (int i = lane; i < 7 * n_obs; i += 64) {
      const double x = live[i];
      smem[i] = x;
      if (A.live_src) live_dev[i] = x;
    }
    for (int i = lane; i < n_obs; i += 64) s_known[i] = rk[i];
  }
  // random vectors the real agent's heuristic uses (best_agent_'s copy)
  const double *rand_g = D.best_rnd + (size_t)pop * 3 * n_obs;

  if (A.do_select) {
    // cost assembly + argmin, B/src/cf_manager.cpp:325-343
    double lmin = 1.7976931348623157e308;
    int lidx = 0x7fffffff;
    for (int a = lane; a < N; a += 64) {
      size_t pa = (size_t)pop * N + a;
      double cost = D.cost_ws[pa];
      double gd = D.goal_dist[pa];
      if (gd > C.approach) cost += gd * CP.k_goal_dist;
      cost += D.path_len[pa] * CP.k_path_len;
      double mo = D.min_obs[pa];
      cost += CP.k_safe_dist / mo;
      if (mo < 2e-5) cost += 10000.0;
      D.costs[pa] = cost;
      s_cost[a] = cost;
      if (cost < lmin) { lmin = cost; lidx = a; }
    }
    group_argmin<64>(lmin, lidx);
    int min_idx = (lidx == 0x7fffffff) ? 0 : lidx;
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

  if (A.do_move) {
    // RealCfAgent::cfPlanner one step, B/src/cf_agent.cpp:343-366
    int gid = A.agent_id ? A.agent_id[pop] : best;
    size_t pg = (size_t)pop * N + gid;
    double k_attr = D.k_attr[pg], k_circ = D.k_circ[pg], k_repel = D.k_repel[pg], k_damp = D.k_damp[pg];
    wave_lds_fence();  // obstacle table + known flags in LDS
    const int ntiles = (M + 63) / 64;
    unsigned long long kb = 0ull;
    V3 F = mk(0.0, 0.0, 0.0);
    if (A.tuned_real_step && ntiles <= 4) {
      double *rrot = D.real_rot + (size_t)pop * 3 * n_obs;
      double *clist = s_cost + N + (N & 1);
      unsigned kb32 = 0u;
      if (ntiles <= 1)
        real_step_w64<1>(D, A.dt_real, pop, lane, htype, smem, s_known, rrot, rand_g, clist, k_attr, k_circ, k_repel,
                         k_damp, goal, init_pos, rp, rv, F, kb32);
      else if (ntiles == 2)
        real_step_w64<2>(D, A.dt_real, pop, lane, htype, smem, s_known, rrot, rand_g, clist, k_attr, k_circ, k_repel,
                         k_damp, goal, init_pos, rp, rv, F, kb32);
      else
        real_step_w64<4>(D, A.dt_real, pop, lane, htype, smem, s_known, rrot, rand_g, clist, k_attr, k_circ, k_repel,
                         k_damp, goal, init_pos, rp, rv, F, kb32);
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
    }
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

  // host-visible outputs first: the caller waits for these only
  if (lane == 0 && A.out) {
    double *o = A.out + pop * 12;
    o[0] = (double)best;
    o[1] = rp.x; o[2] = rp.y; o[3] = rp.z;
    o[4] = rv.x; o[5] = rv.y; o[6] = rv.z;
    o[7] = norm(goal - rp);
    o[8] = rf.x; o[9] = rf.y; o[10] = rf.z;
    if (A.seq != 0.0) {
      __threadfence_system();  // entries 0..10 visible to the host before the sequence number
      *reinterpret_cast<volatile double *>(o + 11) = A.seq;
    }
  }

  if (A.do_reset) {
    // resetEEAgents, B/src/cf_manager.cpp:246-255
    V3 sp, sv;
    if (A.reset_from_real) { sp = rp; sv = rv; }
    else {
      const double *in = A.reset_in + pop * 6;
      sp = mk(in[0], in[1], in[2]);
      sv = mk(in[3], in[4], in[5]);
    }
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
  }
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
  }
  out[i] = r;
}

// winner record {cost, idx, n_points, type, path[cap][3]} per population
__global__ void k_winner(DevView D, double *dst) {
  int pop = blockIdx.x;
  int best = D.best_idx[pop];
  size_t pa = (size_t)pop * D.N + best;
  size_t rec = 4 + (size_t)D.cap * 3;
  double *o = dst + pop * rec;
  if (threadIdx.x == 0) {
    o[0] = D.costs[pa];
    o[1] = (double)best;
    o[2] = (double)D.n_points[pa];
    o[3] = (double)D.types[best];
  }
  const double *path = D.paths + pa * (size_t)D.cap * 3;
  int n3 = D.n_points[pa] * 3;
  for (int i = threadIdx.x; i < D.cap * 3; i += blockDim.x) o[4 + i] = (i < n3) ? path[i] : 0.0;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

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
  int n_blocks = 0;
  size_t lds_rollout = 0, lds_manager = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_mgr = nullptr;
  uint64_t mailbox_seq = 0;     // sequence number of the last pmaf_tick (mailbox entry 11)
  std::vector<void *> allocs;
  std::vector<size_t> alloc_bytes;  // size of every device buffer (state save / load)
  double *h_out = nullptr;      // pinned [P][12] mailbox written by k_manager
  // host copy of the real agent's state (getNextPosition / getNextVelocity /
  // getEEForce / getDistFromGoal must not wait for the running rollout)
  std::vector<double> real_pos_h, real_vel_h, real_force_h;
  double *d_out = nullptr;      // device alias of h_out
  static constexpr int kStage = 4;
  double *h_stage[kStage] = {nullptr, nullptr, nullptr, nullptr};  // pinned staging ring for obstacle SoA uploads
  hipEvent_t ev_stage[kStage] = {nullptr, nullptr, nullptr, nullptr};
  double *h_zc = nullptr, *d_zc = nullptr;  // mapped pinned obstacle buffer read by k_manager in pmaf_tick
  int stage_next = 0;
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

static int pick_lpa(int N, int P, int M) {
  // Heuristic: fill the 1024 SIMDs of the chip with waves, but never use more
  // lanes per agent than there are field obstacles to share.
  // Measured on C2-shaped populations (tools/lpasweep.py, kernel us for 1024 / 2048 / 4096 / 8192 agents):
  //   wave per agent 360 / 549 / 1024 / -,  32 lanes 512 / 492 / 697 / 1348,  16 lanes 672 / 712 / 684 / 1017.
  // The wave-per-agent kernel is the fastest while every wave has a SIMD to itself (<= 1024 waves); a second
  // wave per SIMD costs it more than the group kernels' narrower mapping does, and those run best at 2 per SIMD.
  int lpa = 64;
  while (lpa > 1) {
    long waves = ((long)N * lpa + 63) / 64 * P;
    if (waves > (lpa == 64 ? 1024 : 2048)) lpa /= 2; else break;
  }
  // known-flag bitmask holds 64 tiles per lane
  while ((M + lpa - 1) / lpa > 64 && lpa < 64) lpa *= 2;
  return lpa;
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

static void launch_rollout(pmaf_planner *h) {
  dim3 grid((unsigned)h->n_blocks, (unsigned)h->D.P), block(64);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->profiling) {
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
    h->ev_inflight.emplace_back(e0, e1);
    HIP_CHECK(hipEventRecord(e0, h->stream));
  }
  // (the one-slot kernel keeps lanes 61-63 for the goal and the two speed limits: 62-64 obstacles go to the two-slot kernel)
  const int tiles64 = (h->D.n_obs - 1 >= 62 && h->D.n_obs - 1 <= 64) ? 2 : (h->D.n_obs - 1 + 63) / 64;
  if (h->lpa == 64 && tiles64 <= 4 && !h->force_generic) {
    dim3 g64((unsigned)h->D.N, (unsigned)h->D.P);
#define PMAF_W64(T, F) hipLaunchKernelGGL((k_rollout_w64<T, F>), g64, block, h->lds_rollout, h->stream, h->D, h->cp)
    if (h->math == MATH_FAST) {
      if (tiles64 <= 1) PMAF_W64(1, MATH_FAST); else if (tiles64 == 2) PMAF_W64(2, MATH_FAST); else PMAF_W64(4, MATH_FAST);
    } else if (h->math == MATH_IEEE) {
      if (tiles64 <= 1) PMAF_W64(1, MATH_IEEE); else if (tiles64 == 2) PMAF_W64(2, MATH_IEEE); else PMAF_W64(4, MATH_IEEE);
    } else {
      if (tiles64 <= 1) PMAF_W64(1, MATH_XACT); else if (tiles64 == 2) PMAF_W64(2, MATH_XACT); else PMAF_W64(4, MATH_XACT);
    }
#undef PMAF_W64
  } else if (!h->force_generic && (h->lpa == 32 || h->lpa == 16 || h->lpa == 8) &&
             (h->D.n_obs - 1 + h->lpa - 1) / h->lpa <= 4) {
    const int tl = (h->D.n_obs - 1 + h->lpa - 1) / h->lpa;
#define PMAF_GRP(L, T, M) hipLaunchKernelGGL((k_rollout_grp<L, T, M>), grid, block, h->lds_rollout, h->stream, h->D, h->cp)
#define PMAF_GRP_T(L, M) do { if (tl <= 1) PMAF_GRP(L, 1, M); else if (tl == 2) PMAF_GRP(L, 2, M); else PMAF_GRP(L, 4, M); } while (0)
    // (the opt-in fast arithmetic exists for the w64 kernels only)
    if (h->math == MATH_IEEE) { if (h->lpa == 32) PMAF_GRP_T(32, MATH_IEEE); else if (h->lpa == 16) PMAF_GRP_T(16, MATH_IEEE); else PMAF_GRP_T(8, MATH_IEEE); }
    else { if (h->lpa == 32) PMAF_GRP_T(32, MATH_XACT); else if (h->lpa == 16) PMAF_GRP_T(16, MATH_XACT); else PMAF_GRP_T(8, MATH_XACT); }
#undef PMAF_GRP_T
#undef PMAF_GRP
  } else
  switch (h->lpa) {
#define PMAF_CASE(L) case L: hipLaunchKernelGGL((k_rollout<L>), grid, block, h->lds_rollout, h->stream, h->D, h->cp); break;
    PMAF_CASE(1) PMAF_CASE(2) PMAF_CASE(4) PMAF_CASE(8) PMAF_CASE(16) PMAF_CASE(32) PMAF_CASE(64)
#undef PMAF_CASE
    default: fail(PMAF_ERR_INVALID, "bad lanes_per_agent");
  }
  HIP_CHECK(hipGetLastError());
  if (h->profiling) HIP_CHECK(hipEventRecord(e1, h->stream));
  h->launches++;
  h->scores_valid = true;
  h->cp_valid = true;
  h->rollout_pending = false;
}

static void sync(pmaf_planner *h) {
  HIP_CHECK(hipStreamSynchronize(h->stream));
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
  int total = h->D.P * h->D.N;
  hipLaunchKernelGGL(k_score, dim3((total + 63) / 64), dim3(64), 0, h->stream, h->D, h->cp);
  HIP_CHECK(hipGetLastError());
  h->scores_valid = true;
}

static void upload_live_obstacles(pmaf_planner *h, const double *obstacles) {
  if (!obstacles) return;
  check_range(obstacles, (size_t)h->D.P * h->D.n_obs * 7, "obstacles");
  // ring of pinned staging buffers: wait only for the copy that last used this slot
  int s = h->stage_next;
  h->stage_next = (s + 1) % pmaf_planner::kStage;
  HIP_CHECK(hipEventSynchronize(h->ev_stage[s]));
  aos_to_soa(obstacles, h->h_stage[s], h->D.P, h->D.n_obs);
  HIP_CHECK(hipMemcpyAsync(h->D.obs_live, h->h_stage[s], sizeof(double) * (size_t)h->D.P * 7 * h->D.n_obs,
                           hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(hipEventRecord(h->ev_stage[s], h->stream));
}

// pmaf_tick's obstacles: converted into the mapped pinned buffer k_manager reads directly. One buffer is enough:
// pmaf_tick returns only after the manager kernel that read it has published its result.
static const double *stage_live_obstacles_zero_copy(pmaf_planner *h, const double *obstacles) {
  if (!obstacles) return nullptr;
  check_range(obstacles, (size_t)h->D.P * h->D.n_obs * 7, "obstacles");
  aos_to_soa(obstacles, h->h_zc, h->D.P, h->D.n_obs);
  return h->d_zc;
}

static void launch_manager(pmaf_planner *h, const ManagerArgs &A0) {
  ManagerArgs A = A0;
  A.tuned_real_step = (h->math == MATH_XACT && !h->force_generic) ? 1 : 0;
  hipLaunchKernelGGL(k_manager, dim3((unsigned)h->D.P), dim3(64), h->lds_manager, h->stream, h->D, h->cp, A);
  HIP_CHECK(hipGetLastError());
}

// copy the mailbox into the host-side real-agent state (call after the
// k_manager launch has completed)
static void refresh_real_cache(pmaf_planner *h) {
  for (int p = 0; p < h->D.P; p++) {
    const double *o = h->h_out + p * 12;
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
static void wait_mailbox(pmaf_planner *h, double seq) {
  for (int p = 0; p < h->D.P; p++) {
    const volatile double *s = h->h_out + p * 12 + 11;
    unsigned spins = 0;
    while (*s != seq) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#else
      std::this_thread::yield();
#endif
      if ((++spins & 0x3fffu) == 0) {
        hipError_t e = hipStreamQuery(h->stream);
        if (e == hipErrorNotReady) continue;
        if (e != hipSuccess) throw HipError{e, "hipStreamQuery (mailbox wait)", __LINE__};
        if (*s != seq) fail(PMAF_ERR_DEVICE, "pmaf_tick: manager kernel finished without publishing its result");
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

static void append_real_path(pmaf_planner *h) {
  for (int p = 0; p < h->D.P; p++) {
    const double *o = h->h_out + p * 12;
    h->real_path[p].insert(h->real_path[p].end(), {o[1], o[2], o[3]});
  }
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
    snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s [pmaf_hip.hip:%d]", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
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
    h->math = (prm->flags & PMAF_FLAG_FAST_MATH) ? MATH_FAST : (prm->flags & PMAF_FLAG_IEEE_SEQUENCES) ? MATH_IEEE : MATH_XACT;
    h->lpa = lp ? lp : pick_lpa(N, P, M);
    { const char *fg = getenv("PMAF_FORCE_GENERIC"); h->force_generic = fg && fg[0] == '1'; }
    { const char *ab = getenv("PMAF_ABLATE"); D.ablate = ab ? atoi(ab) : 0; }
    {
      int cus = 0;
      HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
      D.n_simds = 4 * (cus > 0 ? cus : 256);
    }
    REQUIRE((M + h->lpa - 1) / h->lpa <= 64, "pmaf_create: too many obstacles for this lanes_per_agent (need M <= 64*lanes_per_agent)");
    h->n_blocks = (N * h->lpa + 63) / 64;
    {
      // obstacle table + known flags, then (w64 kernels) the per-step list of
      // circular-field terms: 64 * TILES entries of 4 doubles
      size_t off = 7 * (size_t)n_obs + ((size_t)n_obs + 1) / 2;
      off += off & 1;
      // w64: (64 * TILES + 8 padding + 64 scratch) entries; groups: 64 * TILES + one zero entry per group (<= 8)
      h->lds_rollout = sizeof(double) * (off + (64 * 4 + 8 + 64) * 4 + 8 * 4);
    }
    // table | known flags | costs | the tuned real step's list (64 * 4 + 8 + 64 entries of 4 doubles)
    h->lds_manager = sizeof(double) * (7 * (size_t)n_obs + ((size_t)n_obs + 1) / 2 + (size_t)N + 1 + (64 * 4 + 8 + 64) * 4);
    REQUIRE(h->lds_rollout <= 160 * 1024, "pmaf_create: obstacle table does not fit in LDS");

    size_t PN = (size_t)P * N;
    double *goal = h->dalloc<double>(P * 3);
    D.goal = goal;
    D.agent_init_pos = h->dalloc<double>(P * 3);
    D.start_pos = h->dalloc<double>(P * 3);
    D.start_vel = h->dalloc<double>(P * 3);
    D.obs_start = h->dalloc<double>((size_t)P * 7 * n_obs);
    D.known_start = h->dalloc<int32_t>((size_t)P * n_obs);
    D.obs_live = h->dalloc<double>((size_t)P * 7 * n_obs);
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
    HIP_CHECK(hipHostMalloc((void **)&h->h_out, sizeof(double) * P * 12, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h->d_out, h->h_out, 0));
    HIP_CHECK(hipHostMalloc((void **)&h->h_zc, sizeof(double) * (size_t)P * 7 * n_obs, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h->d_zc, h->h_zc, 0));
    for (int i = 0; i < pmaf_planner::kStage; i++) {
      HIP_CHECK(hipHostMalloc((void **)&h->h_stage[i], sizeof(double) * (size_t)P * 7 * n_obs, hipHostMallocDefault));
      HIP_CHECK(hipEventCreateWithFlags(&h->ev_stage[i], hipEventDisableTiming));
    }
    std::memset(h->h_out, 0, sizeof(double) * P * 12);

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
    h->upload(ka, prm->k_attr, PN); h->upload(kc, prm->k_circ, PN);
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
  for (void *p : h->allocs) (void)hipFree(p);
  if (h->h_out) (void)hipHostFree(h->h_out);
  if (h->h_zc) (void)hipHostFree(h->h_zc);
  for (int i = 0; i < pmaf_planner::kStage; i++) {
    if (h->h_stage[i]) (void)hipHostFree(h->h_stage[i]);
    if (h->ev_stage[i]) (void)hipEventDestroy(h->ev_stage[i]);
  }
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
    const int P = D.P, N = D.N, cap = D.cap;
    h->upload(D.agent_init_pos, pos, P * 3);
    h->upload(D.real_init_pos, pos, P * 3);
    h->upload(D.real_pos, pos, P * 3);
    h->upload(D.start_pos, pos, P * 3);
    // CfAgent::setPosition = clear + push_back for every predicted agent
    hipLaunchKernelGGL(k_restart_paths, dim3((P * N + 255) / 256), dim3(256), 0, h->stream, D, D.start_pos);
    HIP_CHECK(hipGetLastError());
    (void)cap;
    h->real_pos_h.assign(pos, pos + P * 3);
    for (int p = 0; p < P; p++)  // RealCfAgent::setPosition = push_back
      h->real_path[p].insert(h->real_path[p].end(), {pos[p * 3], pos[p * 3 + 1], pos[p * 3 + 2]});
    sync(h);
    h->scores_valid = false;
    h->rollout_pending = true;
  });
}

int pmaf_set_real_position(pmaf_planner *h, const double *pos) {
  return guarded([&] {
    REQUIRE(h && pos, "pmaf_set_real_position: NULL argument");
    check_range(pos, (size_t)h->D.P * 3, "pmaf_set_real_position");
    h->use_device();
    sync(h);
    h->upload(h->D.real_pos, pos, h->D.P * 3);
    h->real_pos_h.assign(pos, pos + h->D.P * 3);
    for (int p = 0; p < h->D.P; p++)
      h->real_path[p].insert(h->real_path[p].end(), {pos[p * 3], pos[p * 3 + 1], pos[p * 3 + 2]});
  });
}

int pmaf_start(pmaf_planner *h) {
  return guarded([&] {
    REQUIRE(h, "pmaf_start: NULL handle");
    h->use_device();
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
  });
}

int pmaf_evaluate(pmaf_planner *h, const double *cost_gains, const double *ws, int32_t *best_idx) {
  return guarded([&] {
    REQUIRE(h && cost_gains && ws, "pmaf_evaluate: NULL argument");
    h->use_device();
    set_cost_params(h, cost_gains, ws);
    ensure_scores(h);
    ManagerArgs A{};
    A.do_select = 1;
    A.out = h->d_out;
    launch_manager(h, A);
    sync(h);
    refresh_real_cache(h);
    if (best_idx)
      for (int p = 0; p < h->D.P; p++) best_idx[p] = (int32_t)h->h_out[p * 12];
  });
}

int pmaf_move_real(pmaf_planner *h, const double *obstacles, double dt, int32_t steps, const int32_t *agent_id) {
  return guarded([&] {
    REQUIRE(h && agent_id, "pmaf_move_real: NULL argument");
    REQUIRE(steps >= 0, "pmaf_move_real: steps must be >= 0");
    h->use_device();
    sync(h);
    int32_t hb = 0;
    h->download(&hb, h->D.has_best, 1);
    if (!hb) fail(PMAF_ERR_STATE, "pmaf_move_real: no best agent yet (call pmaf_evaluate first; the reference dereferences a null best_agent_ here)");
    for (int p = 0; p < h->D.P; p++) REQUIRE(agent_id[p] >= 0 && agent_id[p] < h->D.N, "pmaf_move_real: agent_id out of range");
    upload_live_obstacles(h, obstacles);
    h->upload(h->d_agent_id, agent_id, h->D.P);
    for (int s = 0; s < steps; s++) {
      ManagerArgs A{};
      A.do_move = 1;
      A.dt_real = dt;
      A.agent_id = h->d_agent_id;
      A.out = h->d_out;
      launch_manager(h, A);
      sync(h);
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
    sync(h);
    upload_live_obstacles(h, obstacles);
    std::vector<double> in(h->D.P * 6);
    for (int p = 0; p < h->D.P; p++)
      for (int c = 0; c < 3; c++) { in[p * 6 + c] = pos[p * 3 + c]; in[p * 6 + 3 + c] = vel[p * 3 + c]; }
    h->upload(h->d_reset_in, in.data(), in.size());
    ManagerArgs A{};
    A.do_reset = 1;
    A.reset_in = h->d_reset_in;
    A.out = h->d_out;
    launch_manager(h, A);
    sync(h);
    refresh_real_cache(h);
    h->scores_valid = false;
    h->rollout_pending = true;
  });
}

int pmaf_tick(pmaf_planner *h, const double *obstacles, double dt, const double *cost_gains, const double *ws,
              int32_t *best_idx, double *next_pos, double *next_vel) {
  return guarded([&] {
    REQUIRE(h && cost_gains && ws, "pmaf_tick: NULL argument");
    h->use_device();
    set_cost_params(h, cost_gains, ws);
    ensure_scores(h);
    ManagerArgs A{};
    A.live_src = stage_live_obstacles_zero_copy(h, obstacles);
    A.do_select = 1; A.do_move = 1; A.do_reset = 1; A.reset_from_real = 1;
    A.rollout_follows = 1;
    A.dt_real = dt;
    A.out = h->d_out;
    A.seq = (double)(++h->mailbox_seq);
    launch_manager(h, A);
    h->rollout_pending = true;
    launch_rollout(h);
    // outputs of k_manager land in mapped pinned memory; wait for them only
    // (no event between the two launches: the host polls the sequence number)
    wait_mailbox(h, A.seq);
    refresh_real_cache(h);
    append_real_path(h);
    for (int p = 0; p < h->D.P; p++) {
      const double *o = h->h_out + p * 12;
      if (best_idx) best_idx[p] = (int32_t)o[0];
      if (next_pos) { next_pos[p * 3] = o[1]; next_pos[p * 3 + 1] = o[2]; next_pos[p * 3 + 2] = o[3]; }
      if (next_vel) { next_vel[p * 3] = o[4]; next_vel[p * 3 + 1] = o[5]; next_vel[p * 3 + 2] = o[6]; }
    }
  });
}

int pmaf_link_force(pmaf_planner *h, int32_t pop, int32_t n, const double *link_pos, const double *k_r_force,
                    const double *obstacles, double *out) {
  return guarded([&] {
    REQUIRE(h && link_pos && k_r_force && obstacles && out, "pmaf_link_force: NULL argument");
    REQUIRE(pop >= 0 && pop < h->D.P && n >= 0, "pmaf_link_force: bad population or count");
    if (n == 0) return;
    h->use_device();
    double *d_lp = nullptr, *d_k = nullptr, *d_s = nullptr, *d_o = nullptr;
    HIP_CHECK(hipMalloc((void **)&d_lp, sizeof(double) * 3 * n));
    HIP_CHECK(hipMalloc((void **)&d_k, sizeof(double) * n));
    HIP_CHECK(hipMalloc((void **)&d_s, sizeof(double) * 7));
    HIP_CHECK(hipMalloc((void **)&d_o, sizeof(double) * 3 * n));
    const double *sent = obstacles + ((size_t)pop * h->D.n_obs + (h->D.n_obs - 1)) * 7;
    HIP_CHECK(hipMemcpyAsync(d_lp, link_pos, sizeof(double) * 3 * n, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(d_k, k_r_force, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(d_s, sent, sizeof(double) * 7, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_link_force, dim3((n + 63) / 64), dim3(64), 0, h->stream, n, d_lp, d_k, d_s, h->D.C.rad,
                       h->D.C.shell, d_o);
    HIP_CHECK(hipMemcpyAsync(out, d_o, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, h->stream));
    sync(h);
    (void)hipFree(d_lp); (void)hipFree(d_k); (void)hipFree(d_s); (void)hipFree(d_o);
  });
}

#define GETTER_PROLOGUE(name)                    \
  REQUIRE(h, name ": NULL handle");              \
  h->use_device();                               \
  sync(h);                                       \
  const DevView &D = h->D;                       \
  const size_t PN = (size_t)D.P * D.N;           \
  (void)PN;

int pmaf_get_paths(pmaf_planner *h, double *paths, int32_t *n_points) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_paths")
    std::vector<int32_t> np(PN);
    h->download(np.data(), D.n_points, PN);
    if (paths) {
      h->download(paths, D.paths, PN * (size_t)D.cap * 3);
      // entries past an agent's path end are stale device memory: report zeros
      for (size_t pa = 0; pa < PN; pa++)
        std::memset(paths + (pa * D.cap + np[pa]) * 3, 0, sizeof(double) * 3 * (size_t)(D.cap - np[pa]));
    }
    if (n_points) std::memcpy(n_points, np.data(), sizeof(int32_t) * PN);
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
      hb[p] = id[p] > 0;
    }
    h->upload(D.has_best, hb.data(), D.P);
    h->upload(D.best_id, id, D.P);
    h->upload(D.best_type, type, D.P);
    if (rand_vecs) {
      const int n_obs = D.n_obs;
      std::vector<double> r((size_t)D.P * 3 * n_obs);
      for (int p = 0; p < D.P; p++)
        for (int i = 0; i < n_obs; i++)
          for (int c = 0; c < 3; c++) r[((size_t)p * 3 + c) * n_obs + i] = rand_vecs[((size_t)p * n_obs + i) * 3 + c];
      h->upload(D.best_rnd, r.data(), r.size());
    }
  });
}

size_t pmaf_winner_record_doubles(const pmaf_planner *h) { return h ? 4 + (size_t)h->D.cap * 3 : 0; }

int pmaf_write_winner_records(pmaf_planner *h, void *dst_device, size_t bytes) {
  return guarded([&] {
    REQUIRE(h && dst_device, "pmaf_write_winner_records: NULL argument");
    REQUIRE(bytes >= sizeof(double) * pmaf_winner_record_doubles(h) * h->D.P, "pmaf_write_winner_records: buffer too small");
    h->use_device();
    hipLaunchKernelGGL(k_winner, dim3((unsigned)h->D.P), dim3(256), 0, h->stream, h->D, (double *)dst_device);
    HIP_CHECK(hipGetLastError());
  });
}

// ---- checkpoint / resume ----------------------------------------------------
// The blob holds every device buffer of the handle (agents' rotation vectors,
// known flags, paths, real agent, best-agent copy, obstacle tables ...) plus
// the host-side planner state (real agent's trajectory, scoring parameters).
struct StateHeader {
  uint64_t magic;
  int32_t abi, P, N, n_obs, cap, n_bufs;
  int32_t cp_valid, scores_valid, rollout_pending, pad;
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
    char *w = static_cast<char *>(blob);
    StateHeader hd{};
    hd.magic = kStateMagic; hd.abi = PMAF_ABI_VERSION;
    hd.P = h->D.P; hd.N = h->D.N; hd.n_obs = h->D.n_obs; hd.cap = h->D.cap; hd.n_bufs = (int32_t)h->allocs.size();
    hd.cp_valid = h->cp_valid; hd.scores_valid = h->scores_valid; hd.rollout_pending = h->rollout_pending;
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
    std::memcpy(&h->cp, r, sizeof(CostParams)); r += sizeof(CostParams);
    std::memcpy(&h->D.C, r, sizeof(PopConst)); r += sizeof(PopConst);
    for (size_t i = 0; i < h->allocs.size(); i++) {
      HIP_CHECK(hipMemcpyAsync(h->allocs[i], r, h->alloc_bytes[i], hipMemcpyHostToDevice, h->stream));
      r += h->alloc_bytes[i];
    }
    HIP_CHECK(hipStreamSynchronize(h->stream));
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
    h->cp_valid = hd.cp_valid != 0;
    h->scores_valid = hd.scores_valid != 0;
    h->rollout_pending = hd.rollout_pending != 0;
  });
}

void *pmaf_stream(pmaf_planner *h) { return h ? (void *)h->stream : nullptr; }

int pmaf_set_profiling(pmaf_planner *h, int32_t enable) {
  return guarded([&] {
    REQUIRE(h, "pmaf_set_profiling: NULL handle");
    h->use_device();
    sync(h);
    h->profiling = enable != 0;
  });
}
int pmaf_get_kernel_stats(pmaf_planner *h, double *rollout_ms, int64_t *launches, int64_t *agent_steps) {
  return guarded([&] {
    GETTER_PROLOGUE("pmaf_get_kernel_stats")
    if (rollout_ms) *rollout_ms = h->rollout_ms;
    if (launches) *launches = h->profiling ? h->timed_launches : h->launches;
    if (agent_steps) {
      unsigned long long s = 0;
      h->download(&s, D.step_counter, 1);
      *agent_steps = (int64_t)s;
    }
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
    REQUIRE(a && b && out && n > 0 && op >= 0 && op <= 10, "pmaf_debug_math: bad argument");
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_CHECK(hipMalloc((void **)&da, sizeof(double) * n));
    HIP_CHECK(hipMalloc((void **)&db, sizeof(double) * n));
    HIP_CHECK(hipMalloc((void **)&dout, sizeof(double) * n));
    HIP_CHECK(hipMemcpy(da, a, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(db, b, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, da, db, dout);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
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

}  // extern "C"

// pmaf_k_w64.hip -- k_rollout_w64<TILES, MATH>: the wave-per-agent rollout kernel (latency shape: C1, C2, C3) and its
// launcher. Compiled once per arithmetic policy (-DPMAF_W64_MATH=0|1|2|3, csrc/build.sh; 3 with -ffp-contract=fast) so the policies build in
// parallel; each object defines pmaf_k_launch_w64_m<policy>, pmaf_k_launch_w64 (pmaf_k_misc.hip) dispatches.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "pmaf_types.hpp"
#include "pmaf_device.hpp"
#include "pmaf_rollout_w64.hpp"
#include "pmaf_rollout_grp.hpp"

using namespace pmaf;

// ---------------------------------------------------------------------------
// k_rollout_w64<TILES>: one wave64 per agent (see pmaf_rollout_w64.hpp)
// ---------------------------------------------------------------------------
// the step loop, specialised on the agent's heuristic so the per-step code
// carries no type dispatch (the type is uniform per wave)
// SENT: 0 = the repulsive obstacle cannot come into range during this rollout (decided by the caller, one-slot kernel
// only: the step loop then has no block for it), 1 = it can, 2 = decide here at run time (the other kernels)
// PLAIN: every agent of the launch has k_attr != 0 and the agents have unit mass (the reference's defaults and every
// shipped task file; decided by the host at pmaf_create): the step then carries neither the `k_attr != 0` select nor the
// division by the mass -- for a lone wave every instruction is an issue slot
// (ONE LDS copy of exp's data per block: a __shared__ array declared inside rollout_w64_body would be one array per
// instantiation of the body -- six heuristics = 12.4 KB of static LDS per one-wave block, and 2048+ agents x 129 obstacles
// no longer fit eight blocks to a CU: +45 % on those launches, found by tools/regime.py's sweep)
__device__ __forceinline__ double *w64_exp_lds() {
  __shared__ double s_expk[EXPK_N];
  return s_expk;
}
template <int TILES>
__device__ __forceinline__ auto w64_exp_consts(int lane) {
  if constexpr (TILES >= 2) {
    double *tab = w64_exp_lds();
    exp_consts_to_lds(tab, lane);
    wave_lds_fence();
    return exp_consts_from_lds(tab);
  } else {
    return exp_consts_in_vgprs();
  }
}
// STATICV (one slot per lane): every field obstacle of the population is at rest with +0.0 velocities and the lane the
// rel_vel rider needs is idle -- decided per block by the kernel (pmaf_rollout_w64.hpp, NVL)
// SLICE (round 6): launches of MORE waves than the device has SIMDs -- 1 025 ... 2 048 agents on this kernel: two waves on
// half or all of the SIMDs. The issue arbiter serves the older wave first, absolutely: the first n_simds waves run at their
// stand-alone speed T, the others on the slots left over (0.42 of a wave's rate) and then alone, 1.58 T in all (232 -> 375 us
// at 2 048 agents x 32 obstacles, profiles/r6_lpa_band.txt). With SLICE the two waves of a SIMD trade issue priority in
// slices of the wall clock (the group kernel's scheme, pmaf_k_grp.hip) so that both finish together at 2 T / 1.42.
// A template parameter, not a run-time flag: the one-wave-per-SIMD launches (C1, C2, C4) keep their step loop untouched.
template <int TILES, int TYPE, int MATH, int SENT = 2, bool DPPSUM = false, bool PLAIN = false, bool STATICV = false, bool SLICE = false>
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
  // portable_exp's twelve constants: pinned in VGPRs in the one-slot kernel; the multi-slot kernels sit at the
  // 256-VGPR ceiling and fetch them from an LDS table right where the chain uses them (one box, C3: 1151 -> 1128 us
  // against literals / SGPRs; the one-slot kernel the other way round, C2 243 against 264 us: profiles/r3_ab_exp.txt)
  const auto EK = w64_exp_consts<TILES>(lane);
  const size_t pa = (size_t)pop * D.N + a;
  const double *src = D.obs_start + (size_t)pop * 7 * n_obs;
  const int32_t *ks = D.known_start + (size_t)pop * n_obs;
  double *rot_g = D.rot + pa * 3 * n_obs;
  const double *rnd_g = D.rnd + pa * 3 * n_obs;

  LaneObstacles<TILES> O;
  unsigned known_bits = 0u;
#ifndef PMAF_CLOSEST_TABLE
#define PMAF_CLOSEST_TABLE 1
#endif
  // closest-other table (below): the multi-slot kernels' Obstacle / GoalObstacle bodies only -- with one slot per lane
  // the search is one sqrt chain + two reductions, the GoalObstacle agent does not bound the C2 launch without it, and
  // the short C1 rollout would pay for the table's loads in its prologue (measured: profiles/r3_ab_session3.txt)
  constexpr bool USES_CLOSEST = PMAF_CLOSEST_TABLE && TILES >= 2 && (TYPE == T_OBST || TYPE == T_GOALOBST);
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
  // One slot per lane (M <= 60; the host sends 61..64 obstacles to the split / two-slot kernels): the sweep's |ro| /
  // ro.normalized() of the NEXT step are computed at the end of this step, and the lanes that have no obstacle carry
  // the tail's other norms through the same instructions -- lane 63 the goal (distance and direction), lane 62 the
  // speed clamp, lane 61 attractorForce's speed limit -- instead of three more sqrt / reciprocal / divide sequences.
  constexpr bool PRE = (TILES == 1);
  double s_pre = 0.0;
  V3 ron_pre = mk(0.0, 0.0, 0.0);
  // attractorForce's desired velocity (k_attr / k_damp) * g rides in lane 61 as `other * lane_scale`: the lane holds the
  // goal like lane 63 (one-slot kernel; the others use g directly) and lane_scale is the gain ratio there, 1.0 in every
  // other lane (x * 1.0 == x): three multiplies instead of three 64-bit selects per step
  double lane_scale = (lane == 61) ? (k_attr / k_damp) : 1.0;
  asm volatile("" : "+v"(lane_scale));
  if (PRE) {
    // (velocity -0.0 for dt >= 0: p + (-0.0) dt leaves EVERY p as it is, a -0.0 goal coordinate included -- with +0.0
    // it would turn into +0.0, and the goal direction read back from lane 63 also feeds the latch, calc_rot_vec_pre)
    const double nz = (C.dt < 0.0) ? 0.0 : -0.0;
    if (lane == 63 || lane == 61) { O.p[0] = goal; O.v[0] = mk(nz, nz, nz); }
#ifndef PMAF_SENT_RIDER
#define PMAF_SENT_RIDER 1
#endif
    // Round 4 (last session): the loop that carries code for the repulsive obstacle keeps that obstacle in lane 60 like a
    // field obstacle (advanced with the others by predictObstacles' p + v dt), so that |p - sent_pos| and the direction
    // to it -- repelForce's square root, reciprocal and three divisions -- ride in the tail's ONE sequence like the goal
    // and the two speed limits (the one-slot kernel therefore takes at most 60 field obstacles; the host sends 61..64
    // to the other kernels). C4, where the obstacle is the other arm: 280.0 -> 270.5 us per launch
    // (profiles/r4_ab_w64.txt item 8).
    if (PMAF_SENT_RIDER && SENT == 1 && lane == 60) { O.p[0] = sent_p; O.v[0] = sent_v; }
    MT::norm_unit(O.p[0] - p, s_pre, ron_pre);
  }
  constexpr bool SRIDE = PMAF_SENT_RIDER && PRE && SENT == 1;
  V3 verr = attractor_velocity_error<MATH>(v, g, C, k_attr, k_damp);
  const double zsent_lt = D.zsent_lt[pop];
  const bool sent_reachable = (SENT == 2) ? sentinel_reachable(p, sent_p, sent_v, zsent_lt, C, D.cap) : (SENT == 1);
  bool moving = false;  // any field obstacle with a non-zero (or NaN) velocity component
#pragma unroll
  for (int t = 0; t < TILES; t++) moving = moving || !(O.v[t].x == 0.0 && O.v[t].y == 0.0 && O.v[t].z == 0.0);
  moving = wave_any(moving);
  bool advance = true;
  // Closest-other table (Obstacle / GoalObstacle heuristics only; DevView::closest_idx): valid when k_manager computed
  // it for this rollout's start obstacles and they are at rest. The wave keeps its copy in the obstacle-table area of
  // the LDS layout, which this kernel does not use otherwise (its obstacles live in registers).
  const int32_t *cidx = nullptr;
  if (USES_CLOSEST) {
    if (!moving && D.closest_ok[pop] == 1) {   // wave-uniform
      int32_t *s_cidx = reinterpret_cast<int32_t *>(smem);
      const int32_t *ci = D.closest_idx + (size_t)pop * n_obs;
#pragma unroll
      for (int t = 0; t < TILES; t++) {
        const int i = t * 64 + lane;
        if (i < M) s_cidx[i] = ci[i];
      }
      wave_lds_fence();
      cidx = s_cidx;
    }
  }
  V3 repel = mk(0.0, 0.0, 0.0);  // repelForce of the coming step (depends on the step's start state only)
  const double inv_shell = (SENT == 0) ? 0.0 : 1.0 / C.shell;   // (sentinel_repel_m)
  if (sent_reachable) repel = sentinel_repel_m<MATH>(p, C, k_repel, sent_p, sent_r, zsent_lt, inv_shell);
  SecTimers ST;
#ifdef PMAF_SECTION_TIMERS
  ST.start();
#endif
  const bool younger = SLICE && ((((unsigned)blockIdx.y * gridDim.x + blockIdx.x) / (unsigned)D.n_simds) & 1u) != 0u;
  const unsigned slice_log2 = (unsigned)D.prio_slice_log2, younger_of_8 = (unsigned)D.prio_younger_of_8;
  while ((dg > 0.1) && (n < D.cap)) {  // wave-uniform guard, B/src/cf_agent.cpp:310-311
    unsigned long long clk = 0ull;
    if (SLICE) clk = wall_clock64();
    // gate, :315-317
    // |v| < 0.5 vmax and |p - init| < 0.2 on exact squared thresholds
    // (as a lane mask built from single compares: pmaf_rollout_w64.hpp, "lane predicates as masks")
    const lmask gate_m = ~(PMAF_BAL(dg < C.approach) | (PMAF_BAL(zv < C.zvhalf_lt) & PMAF_BAL(z_init < C.zinit_lt)));
    const bool gate = gate_m != 0ull;
    V3 F = mk(0.0, 0.0, 0.0);
    double scale = 1.0;
    PMAF_SEC(ST, 0);
    // (called with the gate closed too: the sweep's few compares then find no obstacle -- one branch less in the step)
#ifdef PMAF_ABLATION   // timing experiments only (PMAF_ABLATE=8: no sweep in the multi-slot kernels)
    if (PRE || (gate && !(D.ablate & 8)))
#else
    if (PRE || gate)
#endif
      circ_and_scale_w64<TILES, TYPE, MATH, PRE, DPPSUM, decltype(EK), (STATICV ? (SENT == 1 ? 59 : 60) : -1)>(lane, p, v, zv, goal, g, dg, gn, C, k_circ, n_obs, rot_g, known_bits,
                                                 O, clist, lane_min, F, scale, ST, EK, D.ablate, 0, s_pre, ron_pre,
                                                 gate_m, cidx);
    PMAF_SEC(ST, 5);
    // attractorForce (:183-193), updatePositionAndVelocity (:253-268)
    // repelForce (:159-181): `repel` was evaluated for this step's start state at the end of the previous step; it is
    // +0.0 when the obstacle cannot come into range, and F -- a sum that started from +0.0 -- is never -0.0, so the
    // unconditional addition is exact (no masked block between the force sum and the tail)
    F = F + (mk(0.0, 0.0, 0.0) + repel);
    // attractorForce (:183-193), updatePositionAndVelocity (:253-258): a = F / mass, |a| <= 13
    V3 acc;
    if (PLAIN) {
      F = F + (scale * k_damp) * verr;
      acc = F;
      const double az = sqn(F);
      if (PMAF_RARE(wave_any(az >= C.zacc_gt))) acc = acc * (13.0 / __builtin_sqrt(az));  // norm(acc) > 13.0
    } else if (PRE) {
      // one slot per lane: as few blocks as possible between the force sum and the tail -- the k_attr == 0 case is a
      // select, and unit mass without clamp (the common case) skips ONE rare block instead of two
      const V3 Fa = F + (scale * k_damp) * verr;
      const bool attr = (k_attr != 0.0);
      F.x = attr ? Fa.x : F.x; F.y = attr ? Fa.y : F.y; F.z = attr ? Fa.z : F.z;
      acc = F;
      double az = sqn(F);
      // (wave-uniform and rare: __any makes the branch a scalar one the block placement can move out of line)
      if (PMAF_RARE(wave_any((C.mass != 1.0) || (az >= C.zacc_gt)))) {
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
    // a -0.0 coordinate into +0.0), so later steps skip it. (Lanes 63 / 61 of the one-slot kernel hold the goal with
    // velocity -0.0: the update leaves it bit for bit.)
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
      const V3 other = PRE ? (O.p[0] - p) : g;   // one-slot kernel: lanes 63 and 61 hold the goal, O.p - p = g there
      const V3 vec = l_nv ? nv : (other * lane_scale);   // lane 61: (k_attr / k_damp) * g = vel_des, the same product
      V3 num = vec;
      num.x = (l_nv || l_des) ? C.vel_max : vec.x;
      // normalized(): the vector itself unless squaredNorm > 0 -- by ONE select on the divisor (x / 1.0 == x for every
      // x, zeros keep their sign, a NaN stays one) instead of three on the quotients; the riders' lanes divide by their
      // norm whatever it is (vel_max / 0 = inf: the fixup of the x component supplies it)
      const double zvec = sqn(vec);
      const double s = MT::sqrt(zvec);
      const double sd = PMAF_LANE(PMAF_BAL(zvec > 0.0) | (3ull << 61)) ? s : 1.0;
      const double rs = MT::rcp_for(sd);
      // (y and z: direction components only; x also carries the riders' vel_max / |.| quotients)
      const V3 q = mk(MT::div_n(num.x, sd, rs), MT::div_n_pos(num.y, sd, rs), MT::div_n_pos(num.z, sd, rs));
      const V3 u = q;
      if (PRE) { s_pre = s; ron_pre = u; }
      const double vn = readlane_d(s, 62), f_nv = readlane_d(q.x, 62), f_des = readlane_d(q.x, 61);
      // (the clamp as a select on the FACTOR: nv * 1.0 is nv exactly -- two selects instead of six)
      v = nv * ((vn > C.vel_max) ? f_nv : 1.0);
      dg = readlane_d(s, 63);
      gn = readlane_v3(u, 63);
      verr = vel_des * smin(1.0, f_des) - v;
      if (SRIDE) {
        // repelForce (:159-181) of the coming step from lane 60's norm and direction: z = |sent_pos - p|^2 is the squared
        // norm of dist_vec (same components up to sign), otr = normalized(p - sent_pos) = -(ro / |ro|) (a zero component's
        // sign differs: F + (0 + repel) below does not see it)
        const double z60 = readlane_d(zvec, 60);
        repel = mk(0.0, 0.0, 0.0);
        if (z60 < zsent_lt) {
          const double s60 = readlane_d(s, 60);
          const V3 u60 = readlane_v3(u, 60);
          double d = s60 - (C.rad + sent_r);
          d = smax(d, 1e-5);
          const V3 otr = -u60;
          const double t = MT::div_pos(1.0, d) - inv_shell;
          const double dd = d * d;
          const V3 num2 = (k_repel * otr) * t;
          repel = MT::div3_n_pos(num2, dd, MT::rcp_for(dd));
        }
      }
    }
    zv = sqn(v);
    z_init = sqn(p - init_pos);
    // every lane stores the (wave-uniform) point to the same address: one transaction, and no exec-masked block
    // in the middle of the tail
    PMAF_BOUND(n < D.cap);
    path[n * 3] = p.x; path[n * 3 + 1] = p.y; path[n * 3 + 2] = p.z;
    n++;
    ran = true;
    if (!PRE && advance) {
#pragma unroll
      for (int t = 0; t < TILES; t++) O.p[t] = O.p[t] + O.v[t] * C.dt;
      advance = moving;
    }
    // (decided at run time in the multi-slot kernels: out of line there -- in the shipped scenes the obstacle sits 170 m
    // away, and a block that is skipped costs a taken branch per step)
    if (SRIDE) {
    } else if ((SENT == 2) ? PMAF_RARE(sent_reachable) : sent_reachable) {  // the only masked block of the step for the repulsive obstacle: advance it, next step's repelForce
      sent_p = sent_p + sent_v * C.dt;
      repel = sentinel_repel_m<MATH>(p, C, k_repel, sent_p, sent_r, zsent_lt, inv_shell);
    }
    PMAF_SEC(ST, 7);
    if (SLICE) {
      if (((((unsigned)(clk >> slice_log2)) & 7u) < younger_of_8) == younger) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
  }
  if (SLICE) __builtin_amdgcn_s_setprio(0);
#ifdef PMAF_SECTION_TIMERS
  if (lane == 0 && pop == 0 && a < 7)
    printf("agent %d type %d steps %d | verr+gate %llu sweep %llu scale %llu circ %llu sum %llu (skip) %llu finish %llu tail %llu | "
           "in-shell steps %llu terms %llu\n", a, TYPE, n - 1, ST.acc[0], ST.acc[1], ST.acc[2], ST.acc[3], ST.acc[4],
           ST.acc[5], ST.acc[6], ST.acc[7], ST.cnt[0], ST.cnt[1]);
#endif

  double cost_ws, path_len;
#ifdef PMAF_SECTION_TIMERS
  const unsigned long long t_loop_end = wall_clock64();
#endif
#ifdef PMAF_ABL_NOCOST     // timing experiments only: no path-cost pass after the loop
  cost_ws = 0.0; path_len = 0.0;
#else
  path_cost_terms_w64<MATH>(lane, path, n, CP.ws, CP.k_workspace, clist, cost_ws, path_len);
#endif
#ifdef PMAF_TICK_STAMPS   // absolute device wall clock (100 MHz) of every wave's start / end: where a tick's time goes
  if (lane == 0 && pop == 0) printf("R %d %llu %llu\n", a, t_begin, wall_clock64());
#endif
#ifdef PMAF_SECTION_TIMERS
  if (lane == 0 && pop == 0 && a < 7)
    printf("agent %d: loop %llu0 ns, path-cost pass %llu0 ns (%d points)\n", a, t_loop_end - t_begin, wall_clock64() - t_loop_end, n);
#endif

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

template <int TILES, int MATH, bool DPPSUM, bool PLAIN, bool SLICE>
__device__ __forceinline__ void rollout_w64_dispatch(const DevView &D, const CostParams &CP) {
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
    // field obstacles at rest with +0.0 velocities (bit patterns: v - (-0.0) would turn a -0.0 component of v into +0.0),
    // and the rider's lane idle (lane 60; 59 when lane 60 holds the repulsive obstacle): the loops with one
    // normalisation sequence less per step
    bool rest = true;
    for (int i = lane; i < M; i += 64) {
      const unsigned long long bx = (unsigned long long)__double_as_longlong(src[3 * n_obs + i]),
                               by = (unsigned long long)__double_as_longlong(src[4 * n_obs + i]),
                               bz = (unsigned long long)__double_as_longlong(src[5 * n_obs + i]);
      rest = rest && ((bx | by | bz) == 0ull);
    }
    const bool st = !wave_any(!rest) && (M <= 59 || !reach);
#define PMAF_BODY(T) \
    if (st) { \
      if (!reach) rollout_w64_body<TILES, T, MATH, 0, DPPSUM, PLAIN, true, SLICE>(D, CP, lane, pop, a); \
      else rollout_w64_body<TILES, T, MATH, 1, DPPSUM, PLAIN, true, SLICE>(D, CP, lane, pop, a); \
    } else if (!reach) rollout_w64_body<TILES, T, MATH, 0, DPPSUM, PLAIN, false, SLICE>(D, CP, lane, pop, a); \
    else rollout_w64_body<TILES, T, MATH, 1, DPPSUM, PLAIN, false, SLICE>(D, CP, lane, pop, a)
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
    case T_GOAL: rollout_w64_body<TILES, T_GOAL, MATH, 2, DPPSUM, PLAIN>(D, CP, lane, pop, a); break;
    case T_OBST: rollout_w64_body<TILES, T_OBST, MATH, 2, DPPSUM, PLAIN>(D, CP, lane, pop, a); break;
    case T_GOALOBST: rollout_w64_body<TILES, T_GOALOBST, MATH, 2, DPPSUM, PLAIN>(D, CP, lane, pop, a); break;
    case T_VEL: rollout_w64_body<TILES, T_VEL, MATH, 2, DPPSUM, PLAIN>(D, CP, lane, pop, a); break;
    case T_RANDOM: rollout_w64_body<TILES, T_RANDOM, MATH, 2, DPPSUM, PLAIN>(D, CP, lane, pop, a); break;
    case T_HAD: rollout_w64_body<TILES, T_HAD, MATH, 2, DPPSUM, PLAIN>(D, CP, lane, pop, a); break;
    default: break;
  }
}

template <int TILES, int MATH, bool DPPSUM, bool PLAIN>
__global__ __launch_bounds__(64) void k_rollout_w64(DevView D, CostParams CP) {
  rollout_w64_dispatch<TILES, MATH, DPPSUM, PLAIN, false>(D, CP);
}
// the one-slot kernel (DPP sum, PLAIN step) with the priority-slicing loop: launches with two waves on a SIMD (SLICE above).
// A kernel of its own so that k_rollout_w64's names -- what the rocprofv3 summaries and profiles/traffic.json are keyed by -- stay.
template <int MATH>
__global__ __launch_bounds__(64) void k_rollout_w64_sliced(DevView D, CostParams CP) {
  rollout_w64_dispatch<1, MATH, true, true, true>(D, CP);
}


#ifndef PMAF_W64_MATH
#error "compile with -DPMAF_W64_MATH=0|1|2|3"
#endif
#define PMAF_CAT2(a, b) a##b
#define PMAF_CAT(a, b) PMAF_CAT2(a, b)
// PMAF_W64_PART (csrc/build.sh, default policy only): the translation unit holds the one-slot kernels (1) or the
// multi-slot kernels (2) alone -- the two are compiled with different pre-RA scheduling directions (measured per
// kernel family: build.sh); undefined = all of them in one unit.
#if defined(PMAF_W64_PART) && PMAF_W64_PART == 1
#define PMAF_W64_LAUNCH PMAF_CAT(PMAF_CAT(pmaf_k_launch_w64_m, PMAF_W64_MATH), _t1)
#elif defined(PMAF_W64_PART) && PMAF_W64_PART == 2
#define PMAF_W64_LAUNCH PMAF_CAT(PMAF_CAT(pmaf_k_launch_w64_m, PMAF_W64_MATH), _tn)
#else
#define PMAF_W64_LAUNCH PMAF_CAT(pmaf_k_launch_w64_m, PMAF_W64_MATH)
#endif
bool PMAF_W64_LAUNCH(const DevView &D, const CostParams &cp, int tiles, bool dppsum,
                     bool plain, size_t lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1, bool slice) {
  const dim3 g64((unsigned)D.N, (unsigned)D.P), block(64);
#define PMAF_L(K) hipExtLaunchKernelGGL(K, g64, block, (unsigned)lds, s, e0, e1, 0, D, cp)
  // one slot per lane: both ordered-sum variants (the host picks); two / four slots: DPP only
#ifdef PMAF_ONLY_W64_1_DPP   // tools/slackprof: a translation unit that holds the C2 kernel alone (same ISA as the product's)
  if (tiles <= 1 && dppsum && plain) { PMAF_L((k_rollout_w64<1, PMAF_W64_MATH, true, true>)); return true; }
  return false;
#endif
#define PMAF_LP(T, S) do { if (plain) PMAF_L((k_rollout_w64<T, PMAF_W64_MATH, S, true>)); \
                           else PMAF_L((k_rollout_w64<T, PMAF_W64_MATH, S, false>)); } while (0)
#if !defined(PMAF_W64_PART) || PMAF_W64_PART == 1
#if PMAF_W64_MATH >= 2
  // two waves per SIMD: the priority-slicing loop (one-slot, DPP sum, PLAIN -- what many-agent populations run)
  if (slice && tiles <= 1 && dppsum && plain) { PMAF_L((k_rollout_w64_sliced<PMAF_W64_MATH>)); return true; }
#endif
  if (tiles <= 1 && !dppsum) { PMAF_LP(1, false); return true; }
  if (tiles <= 1) { PMAF_LP(1, true); return true; }
#endif
#if !defined(PMAF_W64_PART) || PMAF_W64_PART == 2
  if (tiles == 2) { PMAF_LP(2, true); return true; }
  if (tiles > 2 && tiles <= 4) { PMAF_LP(4, true); return true; }
#endif
#undef PMAF_LP
#undef PMAF_L
  return false;
}

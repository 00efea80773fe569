// pmaf_rollout_w64.hpp -- the wave-per-agent rollout step (k_rollout_w64):
// ONE wave64 per agent, lanes over obstacles, TILES = ceil(M/64) obstacle slots
// per lane. This is the latency-bound shape (few agents, e.g. BASELINE C2 = 64
// agents): H sequential agent-steps per wave and nothing to overlap them with,
// so the step is organised around the measured costs of a lone wave on gfx950
// (tools/ubench.hip: dependent FP64 add/fma 6 cycles, IEEE divide 65-72, sqrt
// 110, v_readlane pair + add 40, DPP min stage 49, ds_bpermute stage 98):
//   * the lane's obstacles (position, velocity, radius), their rotation and
//     Random-agent vectors live in REGISTERS for the whole rollout and are
//     advanced there (predictObstacles, B/src/cf_agent.cpp:270-276); the
//     trailing repulsive obstacle is kept wave-uniform in registers;
//   * no barriers / memory fences in the step loop (a __syncthreads() drains
//     the path stores: s_waitcnt vmcnt(0)); path points are stored by every
//     lane to the same address (no exec-masked block) and never waited for;
//   * straight-line blocks so the scheduler can interleave the independent
//     sqrt / divide chains (per-lane circ term computed under predicates);
//   * the tail's three norms (goal distance / direction, speed clamp,
//     attractorForce's speed limit) run as ONE sqrt / reciprocal / divide
//     sequence in lanes 63 / 62 / 61; with one slot per lane the obstacle
//     lanes compute the next step's |ro|, ro.normalized() in that sequence too
//     (rollout_w64_body in pmaf_k_w64.hip);
//   * the sequential `force_ += curr_force` (cf_agent.cpp:106) is reproduced
//     by compacting the non-zero per-obstacle terms, in ascending obstacle
//     index, into an LDS list (v_mbcnt rank) that every lane (lane 0 alone
//     when there are several slots per lane) then sums front to back with
//     broadcast ds_reads -- S dependent adds instead of S x (6 v_readlane +
//     3 adds);
//   * the min-distance, closest-obstacle reductions run as interleaved DPP
//     chains.
// Arithmetic and its order are exactly those of circ_and_scale / finish_step
// in pmaf_device.hpp (bit-identical results; tests/test_parity_gpu.py compares
// every kernel variant with the oracle at zero tolerance).
#pragma once
#include <type_traits>

#include "pmaf_device.hpp"

namespace pmaf {

// finish_step (pmaf_device.hpp) for the w64 kernel: same arithmetic; the
// acceleration clamp is decided on the squared norm (exact threshold) so the
// square root is only taken in the rare clamped case.
// attractorForce's velocity error (B/src/cf_agent.cpp:188-192): depends only on
// the step's start state, so it is evaluated BEFORE the obstacle sweep, where
// its sqrt / divide chain overlaps the sweep's instead of extending the
// dependent chain after the force sum.
template <int MATH>
__device__ __forceinline__ V3 attractor_velocity_error(V3 v, V3 goal_vec, const PopConst &C, double k_attr,
                                                       double k_damp) {
  V3 vel_des = (k_attr / k_damp) * goal_vec;
  double nd, rnd;
  Mth<MATH>::norm_rcp(vel_des, nd, rnd);
  double scale_lim = smin(1.0, Mth<MATH>::div_n(C.vel_max, nd, rnd));
  vel_des = vel_des * scale_lim;
  return vel_des - v;
}

// finish_step (pmaf_device.hpp) for the w64 kernel: same arithmetic and order;
// verr = attractor_velocity_error(...) of this step; the acceleration clamp is
// decided on the squared norm (exact threshold) so its square root is only
// taken in the rare clamped case; the speed clamp is a select so the block is
// not split (the next step's norms can overlap it).
template <int MATH>
__device__ __forceinline__ void finish_step_w64(V3 p, V3 &v, V3 verr, V3 F, double scale, const PopConst &C,
                                                double k_attr, double k_repel, double k_damp, V3 sent_pos,
                                                double sent_rad, double zsent_lt, V3 &new_pos,
                                                const bool sent_reachable = true) {
  if (sent_reachable) {
    V3 ro = sent_pos - p;
    V3 dist_vec = -ro;
    V3 repel = mk(0.0, 0.0, 0.0);
    // max(|dist_vec| - (rad + r), 1e-5) < shell  <=>  |dist_vec|^2 < zsent_lt (host-computed boundary)
    if (sqn(dist_vec) < zsent_lt) {  // rare: strict arithmetic in both modes
      double d = norm(dist_vec) - (C.rad + sent_rad);
      d = smax(d, 1e-5);
      V3 otr = normalized(p - sent_pos);
      double t = 1.0 / d - 1.0 / C.shell;
      double dd = d * d;
      repel = ((k_repel * otr) * t) / dd;
    }
    V3 total = mk(0.0, 0.0, 0.0) + repel;
    F = F + total;
  }
  if (k_attr != 0.0) F = F + (scale * k_damp) * verr;
  V3 acc = F;
  if (C.mass != 1.0) acc = F / C.mass;
  const double az = sqn(acc);
  if (az >= C.zacc_gt) acc = acc * (13.0 / __builtin_sqrt(az));  // norm(acc) > 13.0 (rare)
  V3 half = ((0.5 * acc) * C.dt) * C.dt;
  new_pos = (p + half) + (v * C.dt);
  V3 nv = v + acc * C.dt;
  double vn, rvn;
  Mth<MATH>::norm_rcp(nv, vn, rvn);
  const double f = Mth<MATH>::div_n_pos(C.vel_max, vn, rvn);  // only used when vn > vel_max
  v = nv * ((vn > C.vel_max) ? f : 1.0);   // (select on the factor: nv * 1.0 is nv exactly)
}

// Can the repulsive obstacle come into range at all during this rollout? Per
// step the agent moves at most |a| dt^2 / 2 + |v| dt with |a| <= 13 and
// |v| <= vel_max (both clamped, updatePositionAndVelocity :253-268) and the
// obstacle |v_s| dt, so with the distance at the rollout's start and `steps`
// steps to go the range test of repelForce is decided for the whole rollout
// (margins cover the rounding of the clamps and of this bound). In the shipped
// scenes the obstacle sits 170 m away: the per-step test is skipped.
__device__ __forceinline__ bool sentinel_reachable(V3 p, V3 sent_pos, V3 sent_vel, double zsent_lt, const PopConst &C,
                                                   int steps) {
  const double adt = fabs(C.dt);
  const double per_step = (6.5 * adt * adt + fabs(C.vel_max) * adt) + norm(sent_vel) * adt;
  const double reach = (double)steps * per_step * 1.001 + 1e-9;
  const double range = __builtin_sqrt(zsent_lt) * 1.001;
  const double d0 = norm(p - sent_pos);
  return !(d0 > range + reach);  // NaN anywhere: keep testing
}

// repelForce (B/src/cf_agent.cpp:159-181): only the trailing obstacle repels;
// in range iff |dist_vec|^2 < zsent_lt (host-computed exact boundary of
// max(|dist_vec| - (rad + r), 1e-5) < shell). Rare: strict arithmetic in all
// policies.
__device__ __forceinline__ V3 sentinel_repel(V3 p, const PopConst &C, double k_repel, V3 sent_pos, double sent_rad,
                                             double zsent_lt) {
  const V3 ro = sent_pos - p;
  const V3 dist_vec = -ro;
  V3 repel = mk(0.0, 0.0, 0.0);
  if (sqn(dist_vec) < zsent_lt) {
    double d = norm(dist_vec) - (C.rad + sent_rad);
    d = smax(d, 1e-5);
    const V3 otr = normalized(p - sent_pos);
    const double t = 1.0 / d - 1.0 / C.shell;
    const double dd = d * d;
    repel = ((k_repel * otr) * t) / dd;
  }
  return repel;
}

// sentinel_repel in the kernel's arithmetic policy (round 3): in BASELINE C4 the repulsive obstacle is the other arm's
// end effector and sits inside the range for a good part of every rollout, so the block is not rare there -- with the
// compiler's IEEE expansions (two square roots, seven divisions: ~240 instructions) the step of the C4 kernel was 45 %
// longer than C2's. Same operations on the same operands through the policy's sequences (bit-equal to the IEEE ones in
// the validated range, test_xact_sequences_match_ieee): ONE square root -- |dist_vec| and |p - sent_pos| are the norm of
// the same vector up to sign, so their squared norms are the same double --, normalized() by a select on the divisor,
// 1 / d and the three divisions by d d through refined reciprocals without fixup (d in [1e-5, shell), d d >= 1e-10:
// positive normals; numerators finite); 1 / shell comes in as a loop invariant.
template <int MATH>
__device__ __forceinline__ V3 sentinel_repel_m(V3 p, const PopConst &C, double k_repel, V3 sent_pos, double sent_rad,
                                               double zsent_lt, double inv_shell) {
  typedef Mth<MATH> MT;
  const V3 ro = sent_pos - p;
  const V3 dist_vec = -ro;
  V3 repel = mk(0.0, 0.0, 0.0);
  const double z = sqn(dist_vec);
  if (z < zsent_lt) {
    const double s = MT::sqrt(z);                       // norm(dist_vec) == norm(p - sent_pos)
    double d = s - (C.rad + sent_rad);
    d = smax(d, 1e-5);
    const V3 pms = p - sent_pos;
    const double sd = (z > 0.0) ? s : 1.0;              // normalized(): the vector itself unless squaredNorm > 0
    const V3 otr = MT::div3_n_pos(pms, sd, MT::rcp_for(sd));
    const double t = MT::div_pos(1.0, d) - inv_shell;
    const double dd = d * d;
    const V3 num = (k_repel * otr) * t;
    repel = MT::div3_n_pos(num, dd, MT::rcp_for(dd));
  }
  return repel;
}

// repelForce for the REAL agent's step (RealCfAgent::cfPlanner -> CfAgent::repelForce, B/src/cf_agent.cpp:159-181):
// the live obstacle list is the caller's, so its last radius may differ from the create-time one the host-computed
// squared boundary (zsent_lt) was derived from -- the range test is the reference's own `d < shell` on the live radius.
// Once per tick: the square root does not matter here.
__device__ __forceinline__ V3 sentinel_repel_live(V3 p, const PopConst &C, double k_repel, V3 sent_pos,
                                                  double sent_rad) {
  const V3 ro = sent_pos - p;
  const V3 dist_vec = -ro;
  V3 repel = mk(0.0, 0.0, 0.0);
  double d = norm(dist_vec) - (C.rad + sent_rad);
  d = smax(d, 1e-5);
  if (d < C.shell) {
    const V3 otr = normalized(p - sent_pos);
    const double t = 1.0 / d - 1.0 / C.shell;
    const double dd = d * d;
    repel = ((k_repel * otr) * t) / dd;
  }
  return repel;
}

// wave-level ordering of LDS accesses: DS instructions of one wave execute in
// order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Section timers of the w64 step (debug builds with -DPMAF_SECTION_TIMERS only:
// s_memtime deltas per section, printed by agents 1 and 5 of population 0 at
// the end of their rollout; compiled out otherwise).
#ifdef PMAF_SECTION_TIMERS
struct SecTimers {
  unsigned long long last, acc[8], cnt[4];
  __device__ __forceinline__ void start() { for (int k = 0; k < 8; k++) acc[k] = 0; for (int k = 0; k < 4; k++) cnt[k] = 0; last = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void mark(int k) { unsigned long long now = __builtin_amdgcn_s_memtime(); acc[k] += now - last; last = now; }
};
#define PMAF_SEC(ST, k) (ST).mark(k)
#define PMAF_CNT(ST, k, v) (ST).cnt[k] += (v)
#else
struct SecTimers {};
#define PMAF_SEC(ST, k)
#define PMAF_CNT(ST, k, v)
#endif

// Block layout of the step (round 3): a lone in-order wave pays for every TAKEN branch with an instruction-fetch
// restart, and the compiler lays a conditional block out inline (branch taken to skip it) unless it knows the block is
// cold. The step's rare blocks (first-contact latch, acceleration clamp, lists of more than 16 terms, nothing inside
// the shell) are therefore wave-uniform conditions marked unlikely: the common path falls through.
#ifndef PMAF_EXPECT
#define PMAF_EXPECT 1
#endif
#if PMAF_EXPECT
#define PMAF_RARE(c) __builtin_expect(!!(c), 0)
#else
#define PMAF_RARE(c) (c)
#endif

// Lane predicates as wave-uniform 64-bit MASKS (round 3). The compiler keeps a per-lane bool in an SGPR pair anyway, but a
// vote on a COMPOUND predicate -- ballot(a && b), any(a && !b) -- is lowered through a VGPR (v_cndmask 0/1 + v_cmp_ne_u32:
// two issue slots per vote; only ballot(single compare) folds into the compare). So the step's predicates are built
// from ballots of single compares combined with scalar AND / OR / NOT, tested with a scalar compare, and turned back
// into a per-lane condition with inverse_ballot, which is free (the SGPR pair IS the select's condition operand).
// All of it requires the full wave to be active, which holds wherever these are used (wave-uniform branches only).
typedef unsigned long long lmask;
#define PMAF_BAL(c) __builtin_amdgcn_ballot_w64(c)
#define PMAF_LANE(m) __builtin_amdgcn_inverse_ballot_w64(m)

template <int TILES>
struct LaneObstacles {
  V3 p[TILES], v[TILES];
  double r[TILES];
  double rx[TILES], ry[TILES], rz[TILES];  // field_rotation_vecs_ of this agent
  double qx[TILES], qy[TILES], qz[TILES];  // random_vecs_ of this agent
};

// number of set bits of m below this lane
__device__ __forceinline__ int lane_rank(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ V3 readlane_v3(V3 a, int lane) {
  return mk(readlane_d(a.x, lane), readlane_d(a.y, lane), readlane_d(a.z, lane));
}

// "nearest other field obstacle" of the Obstacle / GoalObstacle heuristics
// (B/src/cf_agent.cpp:434-446 / :480-492: ascending scan, `min_dist > d` with
// min_dist = 100 initially, so the lowest index among the exact minima wins and
// index 0 is the answer when nothing is closer than 100), for every lane that
// latches its slot-`t` obstacle this step: the reference's O(M) scan is done
// by the whole wave on the lanes' register copies of the obstacles (one sqrt
// chain + one DPP argmin per latching lane instead of M dependent sqrt chains).
// Returns, in the latching lanes, the position of that closest obstacle.
template <int TILES, int MATH>
__device__ __forceinline__ V3 closest_other_w64(lmask need_latch_m, int t, int lane, int M,
                                                const LaneObstacles<TILES> &O) {
  V3 cpos = mk(0.0, 0.0, 0.0);
  unsigned long long pend = need_latch_m;
  while (pend) {
    const int L = __ffsll((long long)pend) - 1;  // wave-uniform
    pend &= pend - 1;
    const int id = t * 64 + L;
    V3 own = mk(0.0, 0.0, 0.0);
#pragma unroll
    for (int u = 0; u < TILES; u++)
      if (u == t) own = readlane_v3(O.p[u], L);
    double bd = 100.0;
    int bj = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < TILES; u++) {
      const int j = u * 64 + lane;
      const double d = Mth<MATH>::norm(own - O.p[u]);
      if (j < M && j != id && d < bd) { bd = d; bj = j; }
    }
    const double md = wave_min64(bd);
    const int mj = wave_min64_i((bj != 0x7fffffff && bd == md) ? bj : 0x7fffffff);
    const int c = (mj == 0x7fffffff) ? 0 : mj;
    V3 cp = mk(0.0, 0.0, 0.0);
#pragma unroll
    for (int u = 0; u < TILES; u++)
      if ((c >> 6) == u) cp = readlane_v3(O.p[u], c & 63);
    if (lane == L) cpos = cp;
  }
  return cpos;
}

// closest_other_w64's answer from the closest-other table (DevView::closest_idx, computed by k_manager for rollout-start
// obstacles at rest; `cidx` = the wave's LDS copy): the lane's slot-`t` obstacle looks up the index the reference's scan
// returns and fetches that obstacle's position from the lane that holds it (ds_bpermute; every lane must be active).
// The positions come from the live registers, so the bits are the ones the scan would have read.
template <int TILES>
__device__ __forceinline__ V3 closest_from_table_w64(const int32_t *cidx, int t, int lane, const LaneObstacles<TILES> &O) {
  const int c = cidx[t * 64 + lane];
  V3 cp = mk(0.0, 0.0, 0.0);
#pragma unroll
  for (int u = 0; u < TILES; u++) {
    const V3 q = mk(__shfl(O.p[u].x, c & 63), __shfl(O.p[u].y, c & 63), __shfl(O.p[u].z, c & 63));
    if ((c >> 6) == u) cp = q;
  }
  return cp;
}

// circForce (B/src/cf_agent.cpp:72-108) + attractorForceScaling (:195-227)
// for one agent per wave. clist: LDS, (64*TILES + 8 + 64) entries of 4 doubles
// (the list, 8 entries of zero padding, one scratch entry per lane).
// zv = squaredNorm(v), dg = norm(g), gn = g.normalized() (computed by the caller).
//
// A lone wave issues in order, so a dependent chain costs its full latency
// unless independent instructions stand between its links IN THE SAME BASIC
// BLOCK. After the sweep and the (rare) latches everything is therefore ONE
// straight-line region without branches: the per-lane circular terms (two
// sqrt + divide chains), the closest-obstacle reduction and the wave-uniform
// attractor-scaling chain (sqrt, divide, exp, divide: evaluated speculatively,
// applied iff |F| > 1e-5, :319), the compaction stores (every lane stores: its
// term at its rank, or to its scratch entry) and the first batch of the
// ordered force sum (the list is zero-padded, adding +0.0 is exact).
// PRE: |ro| and ro.normalized() of the lane's slot were computed by the caller (s_pre, ron_pre; one slot per lane
// only) -- the rollout does that at the end of the previous step, see rollout_w64_body.
// DPPSUM: how the ordered force sum is evaluated -- false: LDS list read back in batches of 8 entries (one LDS round
// trip per batch; fewer dependent adds for short lists), true: row-transposed list + DPP row_newbcast chain (one
// ds_read per 16 entries, one dependent add per entry for all three components). Measured crossover at M ~ 20
// field obstacles (tools/msweep.py, 64 agents x 200 steps): M = 16 274 vs 282 us, M = 32 301 vs 287 us, M = 61 353 vs
// 326 us; C3 (M = 128, ~57 terms per step) 1603 -> 1322 us. The host picks per launch (pmaf_host.cpp).
// currentVector's closing `if (cur.norm() < 1e-10) cur = (0,0,1); return cur.normalized()` of the Goal / Velocity branches
// exactly as current_vector<MATH, true> evaluates it (pmaf_device.hpp), with the quotient handed out as well and `keep`
// lanes that take it whatever the norm (a rider of the sequence, see NVL below)
template <int MATH>
__device__ __forceinline__ V3 cv_close_threshold(const V3 cur, const bool keep, V3 &quot) {
  typedef Mth<MATH> M;
  if constexpr (MATH == MATH_XACT) {
    const double z = sqn(cur);
    const double s = M::sqrt_pos(z);
    quot = M::div3_n_pos(cur, s, M::rcp_refined(s));
    const bool tiny = (z < 0x1.79ca10c924223p-67) && !keep;
    return mk(tiny ? 0.0 : quot.x, tiny ? 0.0 : quot.y, tiny ? 1.0 : quot.z);
  } else {
    double s;
    M::template norm_unit<true>(cur, s, quot);
    const bool tiny = (s < 1e-10) && !keep;
    return mk(tiny ? 0.0 : quot.x, tiny ? 0.0 : quot.y, tiny ? 1.0 : quot.z);
  }
}

// NVL (one slot per lane, rollout kernels only; -1: none): the field obstacles are at rest with +0.0 velocities, so
// rel_vel = v - (+0.0) = v EXACTLY in every lane and its normalisation -- one of the step's per-lane sqrt / reciprocal /
// divide sequences -- is the same 32 instructions in all 64 lanes. It rides in idle lane NVL of the current vector's own
// normalisation instead (v goes in for the lane's vector, the quotient is read back): same operations on the same
// operand, one sequence less per step (round 4, last session: C2's Random agents 202.0 -> 190.8 us per rollout).
template <int TILES, int TYPE, int MATH, bool PRE = false, bool DPPSUM = false, class KT = ExpK, int NVL = -1>
__device__ __forceinline__ void circ_and_scale_w64(int lane, V3 p, V3 v, double zv, V3 goal, V3 g, double dg, V3 gn,
                                                   const PopConst &C, double k_circ,
                                                   int n_obs, double *rot_g, unsigned &known_bits,
                                                   LaneObstacles<TILES> &O, double *clist, double &lane_min,
                                                   V3 &F, double &scale, SecTimers &ST, const KT &EK,
                                                   const int ablate = 0, const int rtype = 0,
                                                   const double s_pre = 0.0, const V3 ron_pre = V3{0.0, 0.0, 0.0},
                                                   const lmask gate_m = ~0ull, const int32_t *cidx = nullptr) {
  typedef Mth<MATH> MT;
  // TYPE == T_REAL: the heuristic is a run-time value (the real agent's step in
  // k_manager dispatches to the stored best agent's type, cf_agent.cpp:368-387)
  const int type = (TYPE == T_REAL) ? rtype : TYPE;
  constexpr int BATCH = 8;                 // list entries summed per LDS round trip (10, 12, 16 measured: no gain)
  constexpr int SCRATCH = 64 * TILES + BATCH;
  constexpr int SUM_SCRATCH = (4 * TILES + 1) * 64;  // PMAF_SUM_DPP: 4 TILES chunks of 64 doubles + the padding chunk
  constexpr int MIN_CELL_BOUND = pmaf_list_area_doubles(TILES);   // doubles of the list area (the wave-minimum cell sits behind it)
  (void)MIN_CELL_BOUND;
  (void)SCRATCH; (void)SUM_SCRATCH;
  const int M = n_obs - 1;
  double best_d = C.shell;
  double best_s = 0.0, best_gr = 0.0;  // |ro| and g.ro of the lane's closest obstacle
  int best_i = 0x7fffffff;
  lmask has_best_m = 0ull;             // lanes with best_i != none
  // ---- sweep geometry (circForce :76-88, attractorForceScaling :201-211) ----
  V3 ron_t[TILES], rv_t[TILES];
  double d_t[TILES];
  lmask in_m[TILES];
  lmask any_in_m = 0ull;
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    const int i = t * 64 + lane;
    // lanes whose slot holds a field obstacle (i < M), by scalar arithmetic: loop-invariant, and a ballot of the
    // hoisted compare would cost the VGPR round trip again; gate (:315-317) closed: no obstacle counts
    const int left = M - t * 64;
    const lmask valid_m = gate_m & ((left >= 64) ? ~0ull : ((left <= 0) ? 0ull : ((1ull << left) - 1ull)));
    const V3 ro = O.p[t] - p;
    rv_t[t] = v - O.v[t];
    double s;
    if (PRE) { s = s_pre; ron_t[t] = ron_pre; }
    else MT::template norm_unit<true>(ro, s, ron_t[t]);   // (normalized() by one select on the divisor)
    const lmask skip_m = PMAF_BAL(dot(ron_t[t], gn) < -0.01) & PMAF_BAL(dot(ro, rv_t[t]) < -0.01);
    double d = s - (C.rad + O.r[t]);
    d = smax(d, 1e-5);
    d_t[t] = d;
    const lmask closer_m = valid_m & PMAF_BAL(d < best_d);
    has_best_m |= closer_m;
    const bool closer = PMAF_LANE(closer_m);
    if (TILES == 1) {  // one slot per lane: its |ro| and g.ro are only read from the winning lane
      if (closer) { best_d = d; best_i = i; }
      best_s = s; best_gr = dot(g, ro);
    } else {
      if (closer) { best_d = d; best_i = i; best_s = s; best_gr = dot(g, ro); }
    }
    const lmask live_m = valid_m & ~skip_m;
    if (PMAF_LANE(live_m & PMAF_BAL(d < lane_min))) lane_min = d;
    in_m[t] = live_m & PMAF_BAL(d < C.shell);
    any_in_m |= in_m[t];
  }
  PMAF_SEC(ST, 1);
#ifdef PMAF_ABLATION   // timing experiments only (PMAF_ABLATE=4): no in-shell block
  if (ablate & 4) return;
#endif
  if (PMAF_RARE(any_in_m == 0ull)) return;  // nothing inside the shell: F stays 0, scale stays 1
  PMAF_CNT(ST, 0, 1);
#ifndef PMAF_LDS_MIN
#define PMAF_LDS_MIN 1
#endif
  // Round 3: the wave minimum of the lanes' closest distances (attractorForceScaling's min_dist) through ONE LDS cell:
  // every lane stores the start value, then applies an atomic unsigned-64 minimum with its own distance (non-negative
  // doubles order like their bit patterns; a minimum is exact and order-independent), and the scaling chain reads the
  // cell back -- three DS instructions issued here, their latency under the circular terms, instead of the two 6-stage
  // DPP reductions (12 dependent v_min_u32_dpp + read-backs) in the middle of the step. For a lone wave every
  // instruction is a 4-cycle issue slot (tools/slackprof.py), whatever unit executes it.
  constexpr int MIN_CELL = pmaf_list_area_doubles(TILES);   // the double right behind the list area (pmaf_host.cpp: lds_rollout / lds_manager)
  unsigned long long *min_cell = reinterpret_cast<unsigned long long *>(clist + MIN_CELL);
  // (kernels with long lists only: with the few obstacles of the LDS-batch variant -- C1: nine -- there is not enough
  // work between the atomic and the read to cover the round trip: C1 136.7 -> 140.6 us, measured)
  constexpr bool LDSMIN = PMAF_LDS_MIN && DPPSUM;
  if (LDSMIN) {
    wave_lds_fence();
    *reinterpret_cast<double *>(min_cell) = C.shell;   // every lane, same address, same value
    __hip_atomic_fetch_min(min_cell, (unsigned long long)__double_as_longlong(best_d), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
  }

  // ---- first contact: latch the rotation vector (:92-96, rare) ----
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    const int i = t * 64 + lane;
    const lmask need_latch_m = in_m[t] & PMAF_BAL((known_bits & (1u << t)) == 0u);
    if (PMAF_RARE(need_latch_m != 0ull)) {
      V3 cpos = O.p[t];
      if (type == T_OBST || type == T_GOALOBST) {
        // (cidx: the closest-other table holds for this rollout -- wave-uniform)
        if (cidx) cpos = closest_from_table_w64<TILES>(cidx, t, lane, O);
        else cpos = closest_other_w64<TILES, MATH>(need_latch_m, t, lane, M, O);
      }
      if (PMAF_LANE(need_latch_m)) {
        // (to_obs, the goal vector, its norm and direction are the sweep's / the caller's: calc_rot_vec_pre)
        V3 rot = calc_rot_vec_pre<MATH>(type, p, n_obs, O.p[t], cpos, mk(O.qx[t], O.qy[t], O.qz[t]), ron_t[t], g, dg, gn);
        rot_g[i] = rot.x; rot_g[n_obs + i] = rot.y; rot_g[2 * n_obs + i] = rot.z;
        O.rx[t] = rot.x; O.ry[t] = rot.y; O.rz[t] = rot.z;
        known_bits |= (1u << t);
      }
    }
  }
  PMAF_SEC(ST, 2);
  // ======== straight-line region ========
  // ---- circular-field terms (:97-106), evaluated by every lane (lanes outside
  // the shell compute values that go to their scratch entry)
#ifndef PMAF_MIN_MID
#define PMAF_MIN_MID 1
#endif
  int count = 0;
  double m_mid = 0.0;
  // (two slots only: the four-slot kernel skips slots that hold nothing inside the shell, the last one included)
  constexpr bool MIN_MID = PMAF_MIN_MID && TILES == 2;
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    if (TILES > 2 && in_m[t] == 0ull) continue;  // no term from this slot (wave-uniform; 2 slots: both in one block)
    const V3 rot = mk(O.rx[t], O.ry[t], O.rz[t]);
    const V3 rv = rv_t[t];
    // |rv| is only divided by, and compared with 0: sqrt(z) != 0 <=> z != 0 (for z == 0 the term is discarded, has_c)
    constexpr bool NVR = (NVL >= 0) && PRE && TILES == 1 && TYPE != T_REAL;
    double zrv;
    V3 nv, cur;
    if constexpr (NVR) {
      // (zrv == 0: the quotient is garbage in every lane and every term goes to a scratch entry, has_c below)
      zrv = zv;   // sqn(v): the caller's -- the same expression on the same operand
      const bool rl = (lane == NVL);
      const V3 to_obs = ron_t[t];
      if constexpr (TYPE == T_VEL) {
        // the Velocity heuristic normalises rel_vel itself (nvel): that IS the term's direction, no rider needed
        const V3 nvel = MT::template normalized<true>(rv);
        nv = nvel;
        V3 q;
        cur = cv_close_threshold<MATH>(nvel - to_obs * dot(nvel, to_obs), false, q);
      } else if constexpr (TYPE == T_GOAL) {
        V3 raw = g - to_obs * dot(to_obs, g);
        raw.x = rl ? v.x : raw.x; raw.y = rl ? v.y : raw.y; raw.z = rl ? v.z : raw.z;
        V3 q;
        cur = cv_close_threshold<MATH>(raw, rl, q);
        nv = readlane_v3(q, NVL);
      } else {
        V3 raw = cross(to_obs, rot);
        raw.x = rl ? v.x : raw.x; raw.y = rl ? v.y : raw.y; raw.z = rl ? v.z : raw.z;
        cur = MT::template normalized<true>(raw);
        nv = readlane_v3(cur, NVL);
      }
    } else {
      zrv = sqn(rv);
      double vn, rvn;
      MT::norm_rcp_zpos(zrv, vn, rvn);   // (zrv == 0: garbage that goes to the lane's scratch entry, has_c below)
      nv = MT::div3_n_pos(rv, vn, rvn);
      cur = current_vector<MATH, true>(type, rv, g, ron_t[t], rot);   // (normalized() by a select on the divisor: 4 instructions less)
    }
    // (round 4) multi-slot kernels: the wave minimum's read-back is pinned HERE -- between the last slot's normalisations and
    // its cross products (a scheduling barrier: nothing moves across) -- so that its LDS round trip runs under ~30
    // instructions of arithmetic instead of in front of the closest-obstacle selection (the listing showed ds_read /
    // s_waitcnt lgkmcnt(0) five instructions apart). C3 979.3 -> 963.1 us; the one-slot kernels lose 0.3-0.5 % with it
    // (profiles/r4_ab_w64.txt) and keep the read where it was.
    if (LDSMIN && MIN_MID && t == TILES - 1) {
      wave_lds_fence();
      m_mid = __longlong_as_double((long long)__hip_atomic_load(min_cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef PMAF_ABL_NOCIRC   // timing experiments only (tools/ablate.sh): no circular-term arithmetic, list traffic kept
    const V3 c = rv; (void)nv; (void)cur; (void)rot;
#else
    const V3 c = MT::div_pos(k_circ, d_t[t] * d_t[t]) * unit_triple<MATH>(nv, cur);  // nv x (cur x nv); d >= 1e-5
#endif
    // compact the contributing terms, ascending obstacle index
    const unsigned long long m = in_m[t] & PMAF_BAL(zrv != 0);   // vel_norm != 0, B/src/cf_agent.cpp:98
    const bool has_c = PMAF_LANE(m);
    if (DPPSUM) {
      // row-transposed list: 16 entries per chunk of 64 doubles, [x0..x15 | y0..y15 | z0..z15 | -], so that ONE
      // conflict-free ds_read hands lane 16 r + k component r of entry k (see the sum below); lanes without a term
      // store to a scratch area behind the list
      const int sl = count + lane_rank(m);
      // (the scratch entries use the list's component stride of 16 too -- three blocks of 16 per row of lanes, 192
      // doubles -- so the three stores share one address register and immediate offsets)
      // (list index computed by every lane and SELECTED: left to itself the compiler wraps it into an exec-masked
      // block -- s_and_saveexec / s_or exec and a copy -- for one v_cndmask_b32)
      int li = ((sl >> 4) << 6) + (sl & 15);
      asm("" : "+v"(li));
      const int idx = has_c ? li : (SUM_SCRATCH + (lane >> 4) * 48 + (lane & 15));
      PMAF_BOUND(sl < 64 * TILES && idx + 32 < MIN_CELL_BOUND);
      clist[idx] = c.x; clist[idx + 16] = c.y; clist[idx + 32] = c.z;
    } else {
      const int slot = has_c ? (count + lane_rank(m)) : (SCRATCH + lane);
      PMAF_BOUND(slot * 4 + 2 < MIN_CELL_BOUND);
      double *e = clist + (size_t)slot * 4;
      e[0] = c.x; e[1] = c.y; e[2] = c.z;
    }
    count += __popcll(m);
  }
  if (DPPSUM) {
    // zero padding: the rest of the list's last chunk (the whole next chunk when the list ends on a chunk boundary)
    if ((lane & 15) >= (count & 15)) clist[((count >> 4) << 6) + lane] = 0.0;
  } else {  // zero padding behind the list (8 distinct entries, written by all lanes)
    double *e = clist + (size_t)(count + (lane % BATCH)) * 4;
    e[0] = 0.0; e[1] = 0.0; e[2] = 0.0;
  }
  PMAF_CNT(ST, 1, count);
#ifndef PMAF_SUM_HOIST
#define PMAF_SUM_HOIST 3
#endif
#ifndef PMAF_SUM_FMAC1
#define PMAF_SUM_FMAC1 1
#endif
  // The first chunk of the list is fetched HERE, in front of the attractor-scaling chain (round 3): issued behind the
  // compaction stores (a wave's DS instructions execute in order), its LDS round trip runs under that chain instead
  // of in front of the sum, where the disassembly showed ds_read / s_waitcnt lgkmcnt(0) back to back. (An empty list
  // reads the all-zero padding chunk: the sum below needs no `count > 0` test for it.)
  constexpr bool HOIST1 = DPPSUM && TILES == 1 && (PMAF_SUM_HOIST & 1);
  constexpr bool HOISTN = DPPSUM && TILES >= 2 && (PMAF_SUM_HOIST & 2);
  double e_first = 0.0;
  if (HOIST1 || HOISTN) {
    wave_lds_fence();
    e_first = clist[lane];
  }

  // ---- attractorForceScaling value (:212-226), branchless ----
  double sc;
#ifdef PMAF_ABL_NOSCALE    // timing experiments only: no attractor-scaling chain
  sc = 1.0;
#else
  {
    double m;
    // (round 4: reading the minimum back earlier -- in front of the circular terms -- costs the one-slot kernels 2-3 %
    // (C2 222.3 -> 226.3 us: 12 VGPRs more, another schedule) and gains the two-slot kernel 0.2 %: profiles/r4_ab_w64.txt)
    if (LDSMIN && MIN_MID) {
      m = m_mid;
    } else if (LDSMIN) {
      wave_lds_fence();
      m = __longlong_as_double((long long)__hip_atomic_load(min_cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    } else {
      m = wave_min64(best_d);
    }
    const lmask cand_m = has_best_m & PMAF_BAL(best_d == m);
    int bi;
    if (TILES == 1) {  // obstacle index == lane: lowest candidate lane wins
      bi = cand_m ? (__ffsll((long long)cand_m) - 1) : 0x7fffffff;
    } else {
      // several slots per lane: the lowest obstacle index among the candidates = the lowest candidate lane of the
      // lowest slot that has one (index = slot * 64 + lane) -- scalar bit operations on the candidates' masks per
      // slot instead of a six-stage DPP minimum over the indices
      bi = 0x7fffffff;
#pragma unroll
      for (int t = TILES - 1; t >= 0; t--) {
        const lmask ct = cand_m & PMAF_BAL((best_i >> 6) == t);
        bi = ct ? (t * 64 + __ffsll((long long)ct) - 1) : bi;
      }
    }
    // (the chunk fetched above is not touched before the closest-obstacle reduction is through: the scheduler, left
    // alone, starts the sum ten instructions behind the ds_read and waits for it there)
    if (HOIST1 && (PMAF_SUM_HOIST & 4)) asm("" : "+v"(e_first) : "s"(bi));
    const bool stall = (dot(g, v) <= 0.0) && (zv < C.zv09_lt) && (dg > 0.15);  // norm(v) < vmax - 0.1 vmax
    // (shell == 0: nothing is ever inside it, bi stays "none" and w is discarded)
    // (m in [1e-5, shell) whenever an obstacle is in reach; otherwise bi is "none" and w is discarded)
#ifndef PMAF_EXP_STAGED
#define PMAF_EXP_STAGED 1
#endif
    double w1;
    if (PMAF_EXP_STAGED && TILES >= 2) {   // (constants out of LDS: requested a stage ahead, pmaf_device.hpp)
      typename std::remove_const<KT>::type EKs = EK;
      double m_in = m;
      const ExpHead H = exp_head(EKs, m_in);
      w1 = 1 - portable_exp_nonpos_staged(-MT::div_pos(MT::sqrt_pos(m_in), C.shell), EKs, H);
    }
    ExpPend EP;
    if (!(PMAF_EXP_STAGED && TILES >= 2)) EP = portable_exp_nonpos_begin(-MT::div_pos(MT::sqrt_pos(m), C.shell), EK);
    // |ro| and g.ro of the closest obstacle were computed by the lane that
    // owns it (same operands, same bits as recomputing them here)
    const int bl = bi & 63;
    const double sb = readlane_d(best_s, bl), gr = readlane_d(best_gr, bl);
    double w2 = 1 - MT::div(gr, dg * sb);
    w2 = w2 * w2;
    if (!(PMAF_EXP_STAGED && TILES >= 2)) w1 = 1 - portable_exp_nonpos_end(EP);   // (the table entry has had w2's division to arrive)
    const double w = w1 * w2;
    sc = (bi == 0x7fffffff) ? 1.0 : (stall ? 0.0 : w);
  }
#endif

  // ---- F = ((0 + c_0) + c_1) + ... front to back; every lane reads the same
  // address (LDS broadcast), so every lane ends with the same F. The list is
  // read in batches of BATCH entries (one LDS round trip each).
  // The list is cross-lane communication through LDS: the fences order this
  // lane's reads after (and the next step's writes behind) the other lanes'
  // accesses for the COMPILER -- without them it may reorder or forward its own
  // LDS accesses (observed: wrong sums); the hardware executes a wave's DS
  // instructions in order, so no instruction is emitted for them.
#ifndef PMAF_ABL_NOSUM     // (timing experiments only: without the ordered sum F stays 0)
  if (DPPSUM) {
  // F = ((0 + c_0) + c_1) + ... in ascending obstacle index, WITHOUT an LDS round trip per batch: one ds_read per 16
  // entries puts x_k / y_k / z_k of entry k into lane k of rows 0 / 1 / 2, and one DPP row_newbcast:k move per entry
  // feeds a single v_add_f64 that advances all three component sums at once (row 0 sums x, row 1 y, row 2 z). The
  // moves do not depend on the accumulator, so the dependent chain is ONE add per entry; entries past the end of the
  // list are +0.0 (exact no-op; skipping them in groups of 4 or 8 costs more in branches than the adds: measured).
  if (!(HOIST1 || HOISTN)) wave_lds_fence();
  {
    double acc = 0.0;
    if (TILES >= 2) {
    // Two and four slots per lane (lists of several chunks): move + add fused into ONE instruction per entry: v_fmac_f64_dpp acc += row_newbcast:k(e) * 1.0 -- the product by
    // 1.0 is exact, so the fused multiply-add rounds e + acc exactly like v_add_f64 (the compiler's DPP combiner does
    // not form 64-bit DPP arithmetic, hence the assembly). Hazards the compiler cannot see inside the block: a VALU
    // write of EXEC or of the DPP source ahead of a DPP instruction needs 5 / 2 wait states -> s_nop 4 in front; the
    // block's result is next read by v_readlane / the following block (s_nop 1 behind it covers a DPP reader).
    // Measured (one box): C3 1248.7 -> 1230.9 us, 200 / 256 obstacles 1270 -> 1197 / 1382 -> 1297 us per launch; the
    // one-slot kernel gains nothing (C2 279.1 vs 279.2 us: the block cannot be interleaved with the scaling chain) and
    // keeps the compiler-visible builtin form.
    // Round 3: the next chunk's ds_read is in flight while this chunk is added (one register pair more).
    double one = 1.0;
    asm volatile("" : "+v"(one));
    double e = HOISTN ? e_first : 0.0;
    int c16 = 0;
#ifndef PMAF_SUM_PEELN
#define PMAF_SUM_PEELN 1
#endif
    if (HOISTN && PMAF_SUM_PEELN) {
      // Round 3: the FIRST chunk as sixteen separate statements in the block of the attractor-scaling chain. A fused
      // accumulate needs two issue slots of distance to the next one, so a chunk in one asm block is 16 instructions in
      // 32 slots; the scaling chain (sqrt -> divide -> exp -> divide) is a second dependent chain with the same
      // property and nothing in common with this one -- as separate statements the scheduler can put the one into the
      // other's empty slots. Hazards as in the block below: the DPP source comes out of a ds_read (the s_nop covers a
      // copy the register allocator might place in front), EXEC is only written by SALU instructions on this path.
      double en = clist[(16 << 2) + lane];
      asm volatile("s_nop 1" : "+v"(e));
#define PMAF_FM1(K) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(e), "v"(one));
      PMAF_FM1(0) PMAF_FM1(1) PMAF_FM1(2) PMAF_FM1(3) PMAF_FM1(4) PMAF_FM1(5) PMAF_FM1(6) PMAF_FM1(7)
      PMAF_FM1(8) PMAF_FM1(9) PMAF_FM1(10) PMAF_FM1(11) PMAF_FM1(12) PMAF_FM1(13) PMAF_FM1(14) PMAF_FM1(15)
#undef PMAF_FM1
      asm volatile("s_nop 0" : "+v"(acc));   // (a v_readlane / the next block's DPP reads acc next)
      e = en;
      c16 = 16;
      asm volatile("" : : "v"(sc));          // the scaling value is complete in THIS block (not sunk behind the loop)
    }
    for (; c16 < count; c16 += 16) {
      double en = 0.0;
      if (HOISTN) en = clist[((c16 + 16) << 2) + lane];   // (behind the last chunk: padding / scratch, never used)
      else e = clist[(c16 << 2) + lane];
#define PMAF_FM(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
      asm volatile("s_nop 4\n\t"
                   PMAF_FM(0) PMAF_FM(1) PMAF_FM(2) PMAF_FM(3) PMAF_FM(4) PMAF_FM(5) PMAF_FM(6) PMAF_FM(7)
                   PMAF_FM(8) PMAF_FM(9) PMAF_FM(10) PMAF_FM(11) PMAF_FM(12) PMAF_FM(13) PMAF_FM(14) PMAF_FM(15)
                   "s_nop 1"
                   : "+v"(acc) : "v"(e), "v"(one));
#undef PMAF_FM
      if (HOISTN) e = en;
    }
    } else {
#define PMAF_BC(K) acc = acc + __builtin_amdgcn_update_dpp(e, e, 0x150 + K, 0xf, 0xf, true);
#define PMAF_BC16 PMAF_BC(0) PMAF_BC(1) PMAF_BC(2) PMAF_BC(3) PMAF_BC(4) PMAF_BC(5) PMAF_BC(6) PMAF_BC(7) \
                  PMAF_BC(8) PMAF_BC(9) PMAF_BC(10) PMAF_BC(11) PMAF_BC(12) PMAF_BC(13) PMAF_BC(14) PMAF_BC(15)
    if (HOIST1) {
      // the first chunk unconditionally, in the block of the scaling chain (its 16 dependent adds interleave with that
      // chain's instructions; no branch, no loop for the common list of <= 16 terms); longer lists continue in a loop
#if PMAF_SUM_FMAC1
      {
        // round 3: move + add fused here too (v_fmac_f64_dpp acc += row_newbcast:k(e) * 1.0, exact) -- as SIXTEEN separate
        // statements, so that the scheduler still interleaves them with the scaling chain (the single block of the
        // multi-slot kernels could not be, which is why the fused form did not pay here in round 2). Hazards: the DPP
        // source e comes out of the ds_read, not out of a VALU instruction (2 wait states otherwise; the s_nop in front
        // of the first one covers a copy the register allocator might place there), EXEC is only written by SALU
        // instructions on this path (a VALU write would need 5).
        double one = 1.0;
        asm volatile("" : "+v"(one));
        double e = e_first;
        asm volatile("s_nop 1" : "+v"(e));
#define PMAF_FM1(K) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(e), "v"(one));
        PMAF_FM1(0) PMAF_FM1(1) PMAF_FM1(2) PMAF_FM1(3) PMAF_FM1(4) PMAF_FM1(5) PMAF_FM1(6) PMAF_FM1(7)
        PMAF_FM1(8) PMAF_FM1(9) PMAF_FM1(10) PMAF_FM1(11) PMAF_FM1(12) PMAF_FM1(13) PMAF_FM1(14) PMAF_FM1(15)
#undef PMAF_FM1
      }
#else
      { const double e = e_first; PMAF_BC16 }
#endif
      // (keeps the rest of the scaling chain in THIS block, in front of the rare loop: left alone the compiler sinks
      // it behind the loop, where it can no longer interleave with the 16 dependent adds)
      if (PMAF_SUM_HOIST & 8) asm volatile("" : : "v"(sc), "v"(acc));
      for (int c16 = 16; PMAF_RARE(c16 < count); c16 += 16) {
        const double e = clist[(c16 << 2) + lane];
        PMAF_BC16
      }
    } else {
    for (int c16 = 0; c16 < count; c16 += 16) {
      const double e = clist[(c16 << 2) + lane];
      PMAF_BC16
    }
    }
#undef PMAF_BC16
#undef PMAF_BC
    }
    F = mk(readlane_d(acc, 0), readlane_d(acc, 16), readlane_d(acc, 32));
  }
  wave_lds_fence();
  } else {
  wave_lds_fence();
  // Two or more slots per lane (long lists, C3: ~57 terms = 8 round trips per step): only lane 0 reads and adds --
  // a broadcast ds_read still returns 64 lanes' worth of data, with one active lane the round trip is shorter
  // (C3 1638 -> 1616 us; with one slot per lane the extra branch costs what it saves).
  if (TILES == 1 || lane == 0) {
  for (int k = 0;;) {
    double ex[BATCH], ey[BATCH], ez[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; j++) {
      const double *e = clist + (size_t)(k + j) * 4;
      ex[j] = e[0]; ey[j] = e[1]; ez[j] = e[2];
    }
#pragma unroll
    for (int j = 0; j < BATCH; j++) { F.x = F.x + ex[j]; F.y = F.y + ey[j]; F.z = F.z + ez[j]; }
    k += BATCH;
    if (k >= count) break;
  }
  }
  if (TILES > 1) F = readlane_v3(F, 0);
  wave_lds_fence();
  }
#endif
  PMAF_SEC(ST, 3);
  scale = (sqn(F) >= C.zf_gt) ? sc : scale;  // norm(F) > 1e-5
  PMAF_SEC(ST, 4);
}

// evaluateAgents' path terms (B/src/cf_manager.cpp:302-324 workspace-box
// penalties, :329 / getPathLength cf_agent.cpp:26-32) from the path the rollout
// has just stored. Inside the step loop they cost a dependent sqrt chain and a
// dozen compares per step on a lone wave; here the whole wave works on 64 path
// points at a time: segment norms in parallel, their sum in path order through
// an LDS list (the reference adds them front to back; adding the +0.0 of the
// lanes past the end is exact), the penalties of the (rare) points outside the
// box in path order by lane. Same operands and operations as the in-loop
// evaluation, so the bits are the same.
// The points were written by lane 0: the agent-scope fence + agent-scope loads
// make them visible to the other lanes (not served from a stale L1 line).
// Round 2 measured this pass at ~1.5 us per 64 points (C2: ~6 us of a 300 us launch) and tried to shorten it -- each
// point fetched once with the predecessor moved by DPP wave_shr:1 (halves the pass's fetch traffic), the next block's
// loads in flight during the sum: bit-exact, but C2 303.1 -> 305.7 us / 306.5 us (two variants, A/B on one box,
// tools/ab.sh): the extra live values raise the kernel's SGPR spill count (191 -> 194 / 203) and the step loop pays
// for it. Kept as it was.
__device__ __forceinline__ double ld_agent(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MATH>
__device__ __forceinline__ void path_cost_terms_w64(int lane, const double *path, int n, const double *ws,
                                                    double k_workspace, double *list, double &cost_ws,
                                                    double &path_len) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  cost_ws = 0.0;
  path_len = 0.0;
#ifndef PMAF_COST_PREFETCH
#define PMAF_COST_PREFETCH 1
#endif
  // (round 3: the next block's six loads are in flight while this block is summed -- the pass was one memory round trip
  // per 64 points in front of each sum)
  V3 q_n = mk(0.0, 0.0, 0.0), qp_n = mk(0.0, 0.0, 0.0);
  auto fetch = [&](int base, V3 &q, V3 &qp) __attribute__((always_inline)) {
    const int k = base + lane;
    const bool valid = k < n;
    const bool has_seg = valid && (k > 0);
    const int kk = valid ? k : 0, kp = has_seg ? (k - 1) : 0;
    q = mk(ld_agent(path + kk * 3), ld_agent(path + kk * 3 + 1), ld_agent(path + kk * 3 + 2));
    qp = mk(ld_agent(path + kp * 3), ld_agent(path + kp * 3 + 1), ld_agent(path + kp * 3 + 2));
  };
  if (PMAF_COST_PREFETCH) fetch(0, q_n, qp_n);
  for (int base = 0; base < n; base += 64) {
    const int k = base + lane;
    const bool valid = k < n;
    const bool has_seg = valid && (k > 0);
    V3 q, qp;
    if (PMAF_COST_PREFETCH) {
      q = q_n; qp = qp_n;
      if (base + 64 < n) fetch(base + 64, q_n, qp_n);
    } else {
      fetch(base, q, qp);
    }
    const double seg = Mth<MATH>::norm(q - qp);
    list[lane] = has_seg ? seg : 0.0;
    wave_lds_fence();
#pragma unroll 8
    for (int j = 0; j < 64; j++) path_len += list[j];
    wave_lds_fence();
    const bool out = valid && ((q.x > ws[0]) | (q.x < ws[1]) | (q.y > ws[2]) | (q.y < ws[3]) | (q.z > ws[4]) | (q.z < ws[5]));
    unsigned long long m = wave_ballot(out);
    while (m) {  // rare
      const int L = __ffsll((long long)m) - 1;
      m &= m - 1;
      ws_cost_add(cost_ws, readlane_v3(q, L), ws, k_workspace);
    }
  }
}

}  // namespace pmaf

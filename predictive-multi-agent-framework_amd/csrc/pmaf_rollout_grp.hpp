// pmaf_rollout_grp.hpp -- k_rollout_grp<LPA, TILES>: the throughput shape of
// the rollout (many agents / populations, e.g. BASELINE C5 = 8 x 1024 agents):
// 64/LPA agents per wave64, each evaluated by a group of LPA = 8, 16 or 32
// adjacent lanes that split the obstacle sweep (TILES = ceil(M/LPA) <= 4 slots
// per lane). Same techniques and the same arithmetic as the wave-per-agent
// kernel (pmaf_rollout_w64.hpp): obstacle / rotation / random vectors of the
// lane's slots in registers, no barriers in the step loop, the reference's
// sequential force sum reproduced through a per-group LDS list, group
// reductions by DPP (quad_perm, row_half_mirror, row_mirror) + ds_swizzle.
// Per-agent values are held replicated in the group's lanes; per-agent control
// flow (loop guard, gate, scaling cases, agent type) is predicated, with
// wave-level wave_any() guards around the expensive regions. Against the w64
// kernel this divides the redundantly executed per-agent ("scalar") part and
// the idle lanes of the sweep by 64/LPA, which is what matters once the chip
// is full of waves (FP64-VALU issue bound).
#pragma once
#include "pmaf_device.hpp"
#include "pmaf_rollout_w64.hpp"

namespace pmaf {

// minimum over the LPA lanes of each group, result in every lane of the group
// (fused-DPP u32 stages, see dpp_min_u32; lane ^ 16 by ds_swizzle)
template <int LPA>
__device__ __forceinline__ unsigned group_min_dpp_u32(unsigned v) {
  v = dpp_min_u32<0xB1, 0xf>(v);
  v = dpp_min_u32<0x4E, 0xf>(v);
  v = dpp_min_u32<0x141, 0xf>(v);
  if (LPA >= 16) v = dpp_min_u32<0x140, 0xf>(v);
  if (LPA >= 32) {
    const unsigned o = (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
    v = o < v ? o : v;
  }
  return v;
}
// v >= +0.0 and not NaN in every lane
template <int LPA>
__device__ __forceinline__ double group_min_dpp(double v) {
  const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
  const unsigned H = group_min_dpp_u32<LPA>(hi);
  const unsigned L = group_min_dpp_u32<LPA>((hi == H) ? lo : 0xffffffffu);
  return __hiloint2double((int)H, (int)L);
}
// v >= 0 in every lane
template <int LPA>
__device__ __forceinline__ int group_min_dpp_i(int v) { return (int)group_min_dpp_u32<LPA>((unsigned)v); }

__device__ __forceinline__ V3 shfl_v3(V3 a, int src) {
  return mk(__shfl(a.x, src), __shfl(a.y, src), __shfl(a.z, src));
}

// closest_other_w64 (pmaf_rollout_w64.hpp) for groups of LPA lanes: every
// iteration serves the first latching lane of EACH group (the groups are
// different agents, so their searches are independent); the group's lanes scan
// their register copies of the obstacles, group argmin with lowest-index ties.
template <int LPA, int TILES, int MATH>
__device__ __forceinline__ V3 closest_other_grp(bool srch, int t, int sub, int grp, int M,
                                                const LaneObstacles<TILES> &O) {
  const int lane = grp * LPA + sub;
  const unsigned long long gmask = ((1ull << LPA) - 1ull) << (grp * LPA);
  V3 cpos = mk(0.0, 0.0, 0.0);
  unsigned long long pend = wave_ballot(srch);
  while (pend) {
    const unsigned long long gp = pend & gmask;
    const bool has = gp != 0ull;
    const int L = has ? (__ffsll((long long)gp) - 1) : lane;  // lane (in the wave) served in this group
    const int id = t * LPA + (L - grp * LPA);
    V3 own = mk(0.0, 0.0, 0.0);
#pragma unroll
    for (int u = 0; u < TILES; u++)
      if (u == t) own = shfl_v3(O.p[u], L);
    double bd = 100.0;
    int bj = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < TILES; u++) {
      const int j = u * LPA + sub;
      const double d = Mth<MATH>::norm(own - O.p[u]);
      if (has && j < M && j != id && d < bd) { bd = d; bj = j; }
    }
    const double md = group_min_dpp<LPA>(bd);
    const int mj = group_min_dpp_i<LPA>((bj != 0x7fffffff && bd == md) ? bj : 0x7fffffff);
    const int c = (mj == 0x7fffffff) ? 0 : mj;
    const int csrc = grp * LPA + (c & (LPA - 1));
    V3 cp = mk(0.0, 0.0, 0.0);
#pragma unroll
    for (int u = 0; u < TILES; u++) {
      const V3 q = shfl_v3(O.p[u], csrc);
      if ((c / LPA) == u) cp = q;
    }
    const bool served = has && (lane == L);
    if (served) cpos = cp;
    pend &= ~wave_ballot(served);
  }
  return cpos;
}

// currentVector (pmaf_device.hpp: current_vector) for a wave whose agents may use DIFFERENT heuristics (the group kernel:
// the type is a per-lane run-time value). Evaluated branch by branch a mixed wave -- every population's first two waves
// hold its five heuristic agents, and they bound the C5 launch -- runs one normalisation per branch; here the branches
// only form the un-normalised vector (wave-uniform guards skip a branch no lane needs) and ONE normalisation serves all
// lanes: cur / sqrt(z) with the divisor replaced by 1.0 for z == 0 (normalized() of the Obstacle / GoalObstacle / Random /
// Had branch), and (0, 0, 1) for the Goal / Velocity branch's `cur.norm() < 1e-10` (on the squared norm, see
// current_vector). The Velocity branch's normalized(rel_vel) is the circular term's own nv = rel_vel / |rel_vel| where
// the squared norm is positive, rel_vel itself otherwise. Same operations on the same operands per lane.
template <int MATH>
__device__ __forceinline__ V3 current_vector_grp(int type, V3 rel_vel, V3 nv, double zrv, V3 goal_vec, V3 to_obs, V3 rot) {
  typedef Mth<MATH> M;
  const bool is_goal = (type == T_GOAL), is_vel = (type == T_VEL);
  const lmask goal_m = PMAF_BAL(is_goal), vel_m = PMAF_BAL(is_vel);
  V3 cur = mk(0.0, 0.0, 0.0);
  if (~(goal_m | vel_m) != 0ull) cur = cross(to_obs, rot);
  if (PMAF_RARE(goal_m != 0ull)) {
    const V3 cg = goal_vec - to_obs * dot(to_obs, goal_vec);
    if (is_goal) cur = cg;
  }
  if (PMAF_RARE(vel_m != 0ull)) {
    const V3 nvel = (zrv > 0.0) ? nv : rel_vel;
    const V3 cv = nvel - to_obs * dot(nvel, to_obs);
    if (is_vel) cur = cv;
  }
  const double z = sqn(cur);
  const double s = M::sqrt_pos(z);
  const double sd = (z > 0.0) ? s : 1.0;
  V3 q = M::div3_n_pos(cur, sd, M::rcp_for(sd));
  if (PMAF_RARE((goal_m | vel_m) != 0ull)) {
    if ((is_goal || is_vel) && z < 0x1.79ca10c924223p-67) q = mk(0.0, 0.0, 1.0);
  }
  return q;
}

// circForce + attractorForceScaling for the agents of one wave. `act`: the
// lane's agent takes a step and its gate is open (uniform within the group).
// clist: this GROUP's list in LDS (LPA*TILES entries of 4 doubles + one
// all-zero entry at index LPA*TILES).
// STATIC (round 4): every field obstacle of the wave's population is at rest with velocity components that are +0.0 bit
// for bit (decided per wave in the kernel's prologue). Then rel_vel = v - (+0.0) = v EXACTLY (x - (+0.0) == x for every
// x, -0.0 included), so |rel_vel|, rel_vel / |rel_vel| and the `vel_norm != 0` test (B/src/cf_agent.cpp:97-99) are the
// same for every obstacle of an agent: one sqrt / reciprocal / divide sequence per step (nv_pre, computed by the caller)
// instead of one per slot, and the obstacles' velocities need no registers.
template <int LPA, int TILES, int MATH, bool STATIC = false>
__device__ __forceinline__ void circ_and_scale_grp(lmask act_m, int sub, int grp, int type, V3 p, V3 v, double zv,
                                                   V3 goal, V3 g, double dg, V3 gn, const PopConst &C, double k_circ,
                                                   int n_obs, double *rot_g, unsigned &known_bits,
                                                   LaneObstacles<TILES> &O, double *clist, double &lane_min, V3 &F,
                                                   double &scale, const double *exp_tab,
                                                   const V3 nv_pre = V3{0.0, 0.0, 0.0}) {
  // Lane predicates as scalar masks built from single-compare ballots (pmaf_rollout_w64.hpp, "lane predicates as
  // masks": a vote on a compound predicate costs two VALU instructions, and this kernel is VALU-issue bound).
  typedef Mth<MATH> MT;
  const int M = n_obs - 1;
  // gn = goal_vec.normalized(): the caller's (k_rollout_grp computes it with the tail's other norms)
  double best_d = C.shell;
  double best_s = 0.0, best_gr = 0.0;
  int best_i = 0x7fffffff;
  lmask has_best_m = 0ull;
  int count = 0;
  const unsigned long long gmask = ((1ull << LPA) - 1ull) << (grp * LPA);
  const unsigned long long below = gmask & ((1ull << (grp * LPA + sub)) - 1ull);
  // The group's list is cleared first (round 3): the sum below then reads entry k for every k the longest list of the
  // wave reaches -- a shorter list is followed by +0.0 entries (exact no-ops) -- without the per-entry clamp of the
  // index to an all-zero slot (a compare, a select and an address computation per entry: VALU instructions, which bound
  // this kernel; the clearing is TILES LDS stores per lane with loop-invariant addresses).
  {
    double2 *z = reinterpret_cast<double2 *>(clist) + sub * 2 * TILES;
#pragma unroll
    for (int q = 0; q < 2 * TILES; q++) z[q] = make_double2(0.0, 0.0);
    wave_lds_fence();   // (compiler ordering only: the terms' stores below go to the same entries)
  }
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    const int i = t * LPA + sub;
    const lmask valid_m = act_m & PMAF_BAL(i < M);
    const V3 op = O.p[t];
    const V3 ro = op - p;
    const V3 rv = STATIC ? v : (v - O.v[t]);
    double s;
    V3 ron;
    MT::template norm_unit<true>(ro, s, ron);
    const lmask skip_m = PMAF_BAL(dot(ron, gn) < -0.01) & PMAF_BAL(dot(ro, rv) < -0.01);
    double d = s - (C.rad + O.r[t]);
    d = smax(d, 1e-5);
    const lmask closer_m = valid_m & PMAF_BAL(d < best_d);
    has_best_m |= closer_m;
    if (PMAF_LANE(closer_m)) { best_d = d; best_i = i; best_s = s; best_gr = dot(g, ro); }
    const lmask live_m = valid_m & ~skip_m;
    if (PMAF_LANE(live_m & PMAF_BAL(d < lane_min))) lane_min = d;
    const lmask in_m = live_m & PMAF_BAL(d < C.shell);
    if (in_m != 0ull) {
      const lmask need_latch_m = in_m & PMAF_BAL((known_bits & (1u << t)) == 0u);
      if (PMAF_RARE(need_latch_m != 0ull)) {
        V3 cpos = op;
        const bool need_latch = PMAF_LANE(need_latch_m);
        const bool srch = need_latch && (type == T_OBST || type == T_GOALOBST);
        if (PMAF_RARE(wave_any(srch))) cpos = closest_other_grp<LPA, TILES, MATH>(srch, t, sub, grp, M, O);
        if (need_latch) {
          V3 rot = calc_rot_vec_pre<MATH>(type, p, n_obs, op, cpos, mk(O.qx[t], O.qy[t], O.qz[t]), ron, g, dg, gn);
          rot_g[i] = rot.x; rot_g[n_obs + i] = rot.y; rot_g[2 * n_obs + i] = rot.z;
          O.rx[t] = rot.x; O.ry[t] = rot.y; O.rz[t] = rot.z;
          known_bits |= (1u << t);
        }
      }
      const V3 rot = mk(O.rx[t], O.ry[t], O.rz[t]);
      // |rv| is only divided by and compared with 0 (sqrt(z) != 0 <=> z != 0): no zero / infinity select behind the
      // root, fixup-free divisions -- for z == 0 the term is discarded (has_m)
      double zrv;
      V3 nv;
      if (STATIC) {   // (the caller's: sqn(v) and v / |v| by the same sequence)
        zrv = zv; nv = nv_pre;
      } else {
        zrv = sqn(rv);
        double vn, rvn;
        MT::norm_rcp_zpos(zrv, vn, rvn);
        nv = MT::div3_n_pos(rv, vn, rvn);
      }
#ifndef PMAF_GRP_CURVEC
#define PMAF_GRP_CURVEC 1
#endif
      const V3 cur = PMAF_GRP_CURVEC ? current_vector_grp<MATH>(type, rv, nv, zrv, g, ron, rot)
                                     : current_vector<MATH, true>(type, rv, g, ron, rot);
      const V3 c = MT::div_pos(k_circ, d * d) * unit_triple<MATH>(nv, cur);   // nv x (cur x nv); d >= 1e-5
      const lmask m = in_m & PMAF_BAL(zrv != 0);   // vel_norm != 0, B/src/cf_agent.cpp:98
      if (PMAF_LANE(m)) {
        PMAF_BOUND(count + __popcll(m & below) < LPA * TILES);
        double *e = clist + (size_t)(count + __popcll(m & below)) * 4;
        e[0] = c.x; e[1] = c.y; e[2] = c.z;
      }
      count += __popcll(m & gmask);
    }
  }
  const double m = group_min_dpp<LPA>(best_d);
  if (PMAF_BAL(count > 0) != 0ull) {
    wave_lds_fence();
    // F = ((0 + c_0) + c_1) + ... per group; a group that has run out of terms adds +0.0 (exact no-op)
    // (lanes past their group's last term read the group's all-zero slot)
#ifndef PMAF_GRP_SUM3
#define PMAF_GRP_SUM3 1
#endif
    if (PMAF_GRP_SUM3 && LPA >= 16) {
      // Round 3: the three component sums in three LANES of each DPP row (row lane 0 adds the x of every entry, 1 the y,
      // 2 the z: ONE v_add_f64 per entry instead of three in this VALU-issue-bound kernel; the per-lane ds_read is not a
      // VALU instruction), read back by row_newbcast moves. Same additions in the same order per component. (F enters
      // as +0.0; the other lanes of the row add the entries' padding word.)
      const double *ec = clist + (sub & 3);
      double acc = 0.0;
      for (int k = 0; PMAF_BAL(k < count) != 0ull; k += 4) {   // k + 3 <= LPA * TILES - 1: count <= LPA * TILES
#pragma unroll
        for (int j = 0; j < 4; j++) acc = acc + ec[(size_t)(k + j) * 4];
      }
#define PMAF_RB(x, K) __builtin_amdgcn_update_dpp(x, x, 0x150 + K, 0xf, 0xf, true)
      F = mk(PMAF_RB(acc, 0), PMAF_RB(acc, 1), PMAF_RB(acc, 2));
#undef PMAF_RB
    } else {
    for (int k = 0; PMAF_BAL(k < count) != 0ull; k += 4) {   // k + 3 <= LPA * TILES - 1: count <= LPA * TILES
      const double *e = clist + (size_t)k * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        F.x = F.x + e[j * 4];
        F.y = F.y + e[j * 4 + 1];
        F.z = F.z + e[j * 4 + 2];
      }
    }
    }
    wave_lds_fence();
  }
  const lmask want_m = act_m & PMAF_BAL(sqn(F) >= C.zf_gt);  // norm(F) > 1e-5
  if (want_m != 0ull) {
    const lmask cand_m = has_best_m & PMAF_BAL(best_d == m);
    const int bi = group_min_dpp_i<LPA>(PMAF_LANE(cand_m) ? best_i : 0x7fffffff);
    const int src = grp * LPA + ((bi == 0x7fffffff) ? 0 : (bi & (LPA - 1)));
    // closest obstacle's |ro| and g.ro live in the owning lane's slot registers
    // of tile bi / LPA; its best_* registers hold them iff that lane's own
    // minimum is this obstacle -- true by construction of cand
    const double sb = __shfl(best_s, src), gr = __shfl(best_gr, src);
    double sc = 1.0;
    if (bi == 0x7fffffff) {
      sc = 1;
    } else if (dot(g, v) <= 0.0 && zv < C.zv09_lt && dg > 0.15) {
      sc = 0.0;
    } else {
      const double w1 = 1 - portable_exp_nonpos<MATH>(-MT::div(MT::sqrt(m), C.shell), exp_consts_from_lds(exp_tab));
      double w2 = 1 - MT::div(gr, dg * sb);
      w2 = w2 * w2;
      sc = w1 * w2;
    }
    if (PMAF_LANE(want_m)) scale = sc;
  }
}

// path_cost_terms_w64 (pmaf_rollout_w64.hpp) for groups of LPA lanes: every
// group works on LPA points of ITS agent's path at a time; list = the group's
// LDS list (at least LPA doubles).
template <int LPA, int MATH>
__device__ __forceinline__ void path_cost_terms_grp(int sub, int grp, bool active, const double *path, int n,
                                                    const double *ws, double k_workspace, double *list,
                                                    double &cost_ws, double &path_len) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  cost_ws = 0.0;
  path_len = 0.0;
  const unsigned long long gmask = ((1ull << LPA) - 1ull) << (grp * LPA);
  for (int base = 0; wave_any(active && base < n); base += LPA) {
    const int k = base + sub;
    const bool valid = active && (k < n);
    const bool has_seg = valid && (k > 0);
    const int kk = valid ? k : 0, kp = has_seg ? (k - 1) : 0;
    const V3 q = mk(ld_agent(path + kk * 3), ld_agent(path + kk * 3 + 1), ld_agent(path + kk * 3 + 2));
    const V3 qp = mk(ld_agent(path + kp * 3), ld_agent(path + kp * 3 + 1), ld_agent(path + kp * 3 + 2));
    const double seg = Mth<MATH>::norm(q - qp);
    list[sub] = has_seg ? seg : 0.0;
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < LPA; j++) path_len += list[j];
    wave_lds_fence();
    const bool out = valid && ((q.x > ws[0]) | (q.x < ws[1]) | (q.y > ws[2]) | (q.y < ws[3]) | (q.z > ws[4]) | (q.z < ws[5]));
    unsigned long long gm = wave_ballot(out) & gmask;  // this group's points outside the box, in path order
    while (wave_any(gm != 0ull)) {  // rare
      const int L = gm ? (__ffsll((long long)gm) - 1) : (grp * LPA + sub);
      const V3 ql = shfl_v3(q, L);
      if (gm) ws_cost_add(cost_ws, ql, ws, k_workspace);
      gm &= gm - 1ull;
    }
  }
}

}  // namespace pmaf

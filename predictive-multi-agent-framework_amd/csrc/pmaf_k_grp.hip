// pmaf_k_grp.hip -- k_rollout_grp<LPA, TILES, MATH>: 8/16/32 lanes per agent (throughput shape: C5) and its launcher.
// Compiled once per arithmetic policy (-DPMAF_GRP_MATH=0|2|3; 3 = the contracted policy, with -ffp-contract=fast; the
// plain fast arithmetic, policy 1, exists for the w64 kernels only).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "pmaf_types.hpp"
#include "pmaf_device.hpp"
#include "pmaf_rollout_w64.hpp"
#include "pmaf_rollout_grp.hpp"

using namespace pmaf;

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
//   Re-swept in round 3 on the shorter step (profiles/r3_ab_session3.txt item 11): 6 slices of 8 for the younger wave, slices of
//   2^9 ticks (5 us): 736 us against 761 us with the round-2 setting (5 of 8, 2^10), 746 (6, 2^10), 752 (6, 2^8), 776 (7, 2^10).
#ifndef PMAF_PRIO_SLICE_LOG2
#define PMAF_PRIO_SLICE_LOG2 9
#endif
#ifndef PMAF_PRIO_YOUNGER_OF_8
#define PMAF_PRIO_YOUNGER_OF_8 6
#endif
constexpr int PRIO_SLICE_LOG2 = PMAF_PRIO_SLICE_LOG2;       // 2^9 ticks of the 100 MHz wall clock
constexpr unsigned PRIO_YOUNGER_OF_8 = PMAF_PRIO_YOUNGER_OF_8;
#ifndef PMAF_POP_ROTATE
#define PMAF_POP_ROTATE 8
#endif
constexpr unsigned POP_ROTATE = PMAF_POP_ROTATE;
template <int LPA, int TILES, int MATH, bool STATIC>
__device__ __forceinline__ void rollout_grp_body(const DevView &D, const CostParams &CP) {
  extern __shared__ double smem[];
  __shared__ double s_expk[EXPK_N];   // portable_exp's constants (pmaf_device.hpp: exp_consts_from_lds)
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
  PopConst C = D.C;
#ifndef PMAF_GRP_PIN
#define PMAF_GRP_PIN 1
#endif
  // STATIC body (round 4): the obstacles' velocities no longer hold registers, so the population constants are pinned in
  // VGPRs (as in the wave-per-agent kernel): left in the SGPR file they are spilled to VGPR lanes and reloaded by
  // v_readlane in the step loop -- VALU instructions, which bound this kernel (profiles/r6_c5_strict_steploop.txt)
  if (STATIC && PMAF_GRP_PIN) {
    double *f = reinterpret_cast<double *>(&C);
    for (int i = 0; i < (int)(sizeof(PopConst) / sizeof(double)); i++) asm volatile("" : "+v"(f[i]));
  }
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
    if (STATIC) O.v[t] = mk(0.0, 0.0, 0.0);   // (never read: see circ_and_scale_grp)
    else O.v[t] = mk(src[3 * n_obs + ii], src[4 * n_obs + ii], src[5 * n_obs + ii]);
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
  // (pinning the goal and the start position too changes nothing: 686.1 / 686.7 against 685.6 us, profiles/r4_ab_grp.txt)
  V3 p = mk(D.start_pos[pop * 3], D.start_pos[pop * 3 + 1], D.start_pos[pop * 3 + 2]);
  V3 v = mk(D.start_vel[pop * 3], D.start_vel[pop * 3 + 1], D.start_vel[pop * 3 + 2]);
  const double k_attr = D.k_attr[pa], k_circ = D.k_circ[pa], k_repel = D.k_repel[pa], k_damp = D.k_damp[pa];
  double *path = D.paths + pa * (size_t)D.cap * 3;
  const double zsent_lt = D.zsent_lt[pop];
  const bool sent_reachable = wave_any(sentinel_reachable(p, sent_p, sent_v, zsent_lt, C, D.cap));  // wave-uniform
  bool moving = false;
  if (!STATIC) {
#pragma unroll
    for (int t = 0; t < TILES; t++) moving = moving || !(O.v[t].x == 0.0 && O.v[t].y == 0.0 && O.v[t].z == 0.0);
    moving = wave_any(moving);
  }
  bool advance = !STATIC;   // (STATIC: p + (+0.0) dt is p for every p, -0.0 coordinates included -- never applied)

  int clist_off = 7 * n_obs + (n_obs + 1) / 2;
  clist_off += clist_off & 1;
  double *clist = smem + clist_off + (size_t)grp * ((LPA * TILES + 1) * 4);
  if (sub < 4) clist[(size_t)LPA * TILES * 4 + sub] = 0.0;  // the group's all-zero list entry
  exp_consts_to_lds(s_expk, lane);
  wave_lds_fence();

  double lane_min = C.shell;
  int n = 1;
  bool ran = false;
  if (active && sub == 0) { path[0] = p.x; path[1] = p.y; path[2] = p.z; }

  V3 g = goal - p;
  double dg = Mth<MATH>::norm(g);
  double zv = sqn(v);
  double z_init = sqn(p - init_pos);
  // goal_vec.normalized() (x / 1.0 == x) and attractorForce's velocity error of the coming step: loop-carried, see the
  // packed tail below
  typedef Mth<MATH> MT;
  V3 gn = MT::div3(g, (dg > 0.0) ? dg : 1.0);
  V3 verr = attractor_velocity_error<MATH>(v, g, C, k_attr, k_damp);
  // Packed tail (round 3, 16 and 32 lanes per agent: a group is one or two DPP rows). The step needs three norms of
  // per-agent vectors -- |g| with g.normalized(), |nv| with vel_max / |nv| (speed clamp), vel_max / |vel_des|
  // (attractorForce's limit) -- which every lane of the group used to evaluate as separate sqrt / reciprocal / divide
  // sequences (~110 VALU instructions per step of a VALU-issue-bound kernel). Lanes 0 / 1 / 2 of each row now carry the
  // three vectors through ONE sequence (the w64 kernel's idle-lane riders) and the results come back by DPP
  // row_newbcast moves: the same operations on the same operands, so the same bits.
#ifndef PMAF_GRP_PACK
#define PMAF_GRP_PACK 1
#endif
  constexpr bool PACK = PMAF_GRP_PACK && (LPA >= 16);
  const double inv_shell = 1.0 / C.shell;   // (sentinel_repel_m)
  const int rsub = sub & 15;
  const bool l_nv = (rsub == 1), l_des = (rsub == 2);
  double lane_scale = l_des ? (k_attr / k_damp) : 1.0;   // vel_des = (k_attr / k_damp) * g rides as g * lane_scale
  asm volatile("" : "+v"(lane_scale));
  const unsigned lin_block = blockIdx.y * gridDim.x + blockIdx.x;
  const bool shared_simd = gridDim.x * gridDim.y > (unsigned)D.n_simds;
  const bool younger = ((lin_block / (unsigned)D.n_simds) & 1u) != 0u;
  const lmask active_m = PMAF_BAL(a < D.N);   // (outside the loop: a ballot of a hoisted compare would go through a VGPR)
  while (true) {
    // B/src/cf_agent.cpp:310-311, per agent -- as a lane mask built from single compares (pmaf_rollout_w64.hpp)
    const lmask run_m = active_m & PMAF_BAL(dg > 0.1) & PMAF_BAL(n < D.cap);
    if (run_m == 0ull) break;
    const bool run = PMAF_LANE(run_m);
    unsigned long long clk = 0ull;
    if (shared_simd) clk = wall_clock64();
    // gate, :315-317
    const lmask gate_m = ~(PMAF_BAL(dg < C.approach) | (PMAF_BAL(zv < C.zvhalf_lt) & PMAF_BAL(z_init < C.zinit_lt)));
    if (!PACK) verr = attractor_velocity_error<MATH>(v, g, C, k_attr, k_damp);
    V3 F = mk(0.0, 0.0, 0.0);
    double scale = 1.0;
    if ((run_m & gate_m) != 0ull) {
      V3 nv_pre = mk(0.0, 0.0, 0.0);
      if (STATIC) {   // rel_vel / |rel_vel| of every obstacle of this agent (garbage for zv == 0: the terms are discarded then)
        double vn0, rvn0;
        MT::norm_rcp_zpos(zv, vn0, rvn0);
        nv_pre = MT::div3_n_pos(v, vn0, rvn0);
      }
      circ_and_scale_grp<LPA, TILES, MATH, STATIC>(run_m & gate_m, sub, grp, type, p, v, zv, goal, g, dg, gn, C, k_circ, n_obs,
                                     rot_g, known_bits, O, clist, lane_min, F, scale, s_expk, nv_pre);
    }
    V3 new_pos;
    V3 nv = v;
    if (PACK) {
      // finish_step_w64 up to the new velocity (repelForce :159-181, attractorForce :183-193, the acceleration clamp and
      // the integration :253-258) ...
      if (sent_reachable) F = F + (mk(0.0, 0.0, 0.0) + sentinel_repel_m<MATH>(p, C, k_repel, sent_p, sent_r, zsent_lt, inv_shell));
      if (k_attr != 0.0) F = F + (scale * k_damp) * verr;
      V3 acc = F;
      if (C.mass != 1.0) acc = F / C.mass;
      const double az = sqn(acc);
      if (az >= C.zacc_gt) acc = acc * (13.0 / __builtin_sqrt(az));  // norm(acc) > 13.0 (rare)
      const V3 half = ((0.5 * acc) * C.dt) * C.dt;
      new_pos = (p + half) + (v * C.dt);
      nv = v + acc * C.dt;
    } else {
      finish_step_w64<MATH>(p, nv, verr, F, scale, C, k_attr, k_repel, k_damp, sent_p, sent_r, zsent_lt, new_pos,
                            sent_reachable);
    }
    if (run) {
      p = new_pos;
      g = goal - p;
      if (PACK) {
        // ... and the three norms in one sequence: row lane 0 the goal vector, 1 the new velocity, 2 the desired velocity
        const V3 vel_des = (k_attr / k_damp) * g;
        const V3 other = g * lane_scale;
        const V3 vec = mk(l_nv ? nv.x : other.x, l_nv ? nv.y : other.y, l_nv ? nv.z : other.z);
        const double numx = (l_nv || l_des) ? C.vel_max : vec.x;
        const double zvec = sqn(vec);
        const double s = MT::sqrt(zvec);
        // normalized(): the vector itself unless squaredNorm > 0, by a select on the divisor (x / 1.0 == x); the riders
        // divide vel_max by their norm whatever it is (vel_max / 0 = inf: the fixup of the x component supplies it)
        const double sd = ((zvec > 0.0) || l_nv || l_des) ? s : 1.0;
        const double rs = MT::rcp_for(sd);
        const V3 q = mk(MT::div_n(numx, sd, rs), MT::div_n_pos(vec.y, sd, rs), MT::div_n_pos(vec.z, sd, rs));
#define PMAF_RB(x, K) __builtin_amdgcn_update_dpp(x, x, 0x150 + K, 0xf, 0xf, true)
        dg = PMAF_RB(s, 0);
        gn = mk(PMAF_RB(q.x, 0), PMAF_RB(q.y, 0), PMAF_RB(q.z, 0));
        const double vn = PMAF_RB(s, 1), f_nv = PMAF_RB(q.x, 1), f_des = PMAF_RB(q.x, 2);
#undef PMAF_RB
        v = nv * ((vn > C.vel_max) ? f_nv : 1.0);   // (select on the factor: nv * 1.0 is nv exactly)
        verr = vel_des * smin(1.0, f_des) - v;
      } else {
        v = nv;
        dg = MT::norm(g);
        gn = MT::div3(g, (dg > 0.0) ? dg : 1.0);
      }
      zv = sqn(v);
      z_init = sqn(p - init_pos);
      PMAF_BOUND(n < D.cap);
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


// the kernel: per wave, are the population's field obstacles at rest with +0.0 velocities (bit patterns)? Then the
// STATIC body (one rel_vel normalisation per step, no velocity registers); otherwise the general one.
template <int LPA, int TILES, int MATH>
__global__ __launch_bounds__(64) void k_rollout_grp(DevView D, CostParams CP) {
  const int n_obs = D.n_obs, M = n_obs - 1;
  const double *src = D.obs_start + (size_t)blockIdx.y * 7 * n_obs;
  bool rest = true;
  for (int i = threadIdx.x; i < M; i += 64) {
    const unsigned long long bx = (unsigned long long)__double_as_longlong(src[3 * n_obs + i]),
                             by = (unsigned long long)__double_as_longlong(src[4 * n_obs + i]),
                             bz = (unsigned long long)__double_as_longlong(src[5 * n_obs + i]);
    rest = rest && ((bx | by | bz) == 0ull);
  }
#ifdef PMAF_GRP_FORCE_STATIC   // register-budget experiments only: the STATIC body alone
  (void)rest;
  rollout_grp_body<LPA, TILES, MATH, true>(D, CP);
#else
  if (!wave_any(!rest)) rollout_grp_body<LPA, TILES, MATH, true>(D, CP);
  else rollout_grp_body<LPA, TILES, MATH, false>(D, CP);
#endif
}

#ifndef PMAF_GRP_MATH
#error "compile with -DPMAF_GRP_MATH=0|2|3"
#endif
#define PMAF_CAT2(a, b) a##b
#define PMAF_CAT(a, b) PMAF_CAT2(a, b)
bool PMAF_CAT(pmaf_k_launch_grp_m, PMAF_GRP_MATH)(const DevView &D, const CostParams &cp, int lpa, int tiles,
                                                  int n_blocks, size_t lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const dim3 grid((unsigned)n_blocks, (unsigned)D.P), block(64);
#define PMAF_GRP(L, T) hipExtLaunchKernelGGL((k_rollout_grp<L, T, PMAF_GRP_MATH>), grid, block, (unsigned)lds, s, e0, e1, 0, D, cp)
#define PMAF_GRP_T(L) do { if (tiles <= 1) PMAF_GRP(L, 1); else if (tiles == 2) PMAF_GRP(L, 2); else PMAF_GRP(L, 4); } while (0)
  if (tiles > 4) return false;
  if (lpa == 32) PMAF_GRP_T(32);
  else if (lpa == 16) PMAF_GRP_T(16);
  else if (lpa == 8) PMAF_GRP_T(8);
  else return false;
#undef PMAF_GRP_T
#undef PMAF_GRP
  return true;
}

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
//     the path stores: s_waitcnt vmcnt(0)); path points are stored by lane 0
//     and never waited for;
//   * straight-line blocks so the scheduler can interleave the independent
//     sqrt / divide chains (per-lane circ term computed under predicates, the
//     next step's goal distance / speed / start distance norms are computed
//     together with this step's path-length norm);
//   * the sequential `force_ += curr_force` (cf_agent.cpp:106) is reproduced
//     by compacting the non-zero per-obstacle terms, in ascending obstacle
//     index, into an LDS list (v_mbcnt rank) that every lane then sums
//     front to back with broadcast ds_reads -- S dependent adds instead of
//     S x (6 v_readlane + 3 adds);
//   * the min-distance, closest-obstacle reductions run as interleaved DPP
//     chains.
// Arithmetic and its order are exactly those of circ_and_scale / finish_step
// in pmaf_device.hpp (bit-identical results; tests/test_parity_gpu.py compares
// every kernel variant with the oracle at zero tolerance).
#pragma once
#include "pmaf_device.hpp"

namespace pmaf {

// wave-level ordering of LDS accesses: DS instructions of one wave execute in
// order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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

// circForce (B/src/cf_agent.cpp:72-108) + attractorForceScaling (:195-227)
// for one agent per wave. clist: LDS, 64*TILES entries of 4 doubles.
// nrm_v = norm(v), dg = norm(g) (already computed by the caller).
template <int TILES, int TYPE>
__device__ __forceinline__ void circ_and_scale_w64(int lane, V3 p, V3 v, double nrm_v, V3 goal, V3 g, double dg,
                                                   const PopConst &C, double k_circ, const ObsTab &T,
                                                   int n_obs, double *rot_g, unsigned &known_bits,
                                                   LaneObstacles<TILES> &O, double *clist, double &min_obs,
                                                   V3 &F, double &scale) {
  const int M = n_obs - 1;
  // goal_vec.normalized(): dg == sqrt(squaredNorm(g)), the value normalized() divides by
  const V3 gn = (dg > 0.0) ? (g / dg) : g;
  double lane_min = min_obs;
  double best_d = C.shell;
  int best_i = 0x7fffffff;
  int count = 0;
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    const int i = t * 64 + lane;
    const bool valid = i < M;
    const V3 op = O.p[t];
    const V3 ro = op - p;
    const V3 rv = v - O.v[t];
    const double z = sqn(ro);
    const double s = __builtin_sqrt(z);
    const V3 ron = (z > 0.0) ? (ro / s) : ro;
    const bool skip = (dot(ron, gn) < -0.01) && (dot(ro, rv) < -0.01);
    double d = s - (C.rad + O.r[t]);
    d = smax(d, 1e-5);
    if (valid && d < best_d) { best_d = d; best_i = i; }
    const bool live = valid && !skip;
    if (live && d < lane_min) lane_min = d;
    const bool in_shell = live && (d < C.shell);
    if (__any(in_shell)) {
      // first contact: latch the rotation vector (rare)
      const bool need_latch = in_shell && !((known_bits >> t) & 1u);
      if (__any(need_latch)) {
        if (need_latch) {
          V3 rot = calc_rot_vec(TYPE, p, goal, T, n_obs, i, op, mk(O.qx[t], O.qy[t], O.qz[t]));
          rot_g[i] = rot.x; rot_g[n_obs + i] = rot.y; rot_g[2 * n_obs + i] = rot.z;
          O.rx[t] = rot.x; O.ry[t] = rot.y; O.rz[t] = rot.z;
          known_bits |= (1u << t);
        }
      }
      // per-lane circular-field term, evaluated by every lane (lanes outside
      // the shell compute values that are discarded by has_c)
      const V3 rot = mk(O.rx[t], O.ry[t], O.rz[t]);
      const double vn = norm(rv);
      const V3 nv = rv / vn;
      const V3 cur = current_vector(TYPE, rv, g, ron, rot);
      const V3 c = (k_circ / (d * d)) * cross(nv, cross(cur, nv));
      const bool has_c = in_shell && (vn != 0);
      // compact the contributing terms, ascending obstacle index, into LDS
      const unsigned long long m = __ballot(has_c);
      if (has_c) {
        double *e = clist + (size_t)(count + lane_rank(m)) * 4;
        e[0] = c.x; e[1] = c.y; e[2] = c.z;
      }
      count += __popcll(m);
    }
  }
  // interleaved DPP reductions
  const double mo = wave_min64(lane_min);
  const double m = wave_min64(best_d);
  min_obs = mo;
  if (count > 0) {
    wave_lds_fence();
    // F = ((0 + c_0) + c_1) + ... front to back; every lane reads the same
    // address (LDS broadcast), so every lane ends with the same F
    int k = 0;
    for (; k + 4 <= count; k += 4) {
      const double *e = clist + (size_t)k * 4;
      V3 c0 = mk(e[0], e[1], e[2]), c1 = mk(e[4], e[5], e[6]), c2 = mk(e[8], e[9], e[10]), c3 = mk(e[12], e[13], e[14]);
      F = F + c0; F = F + c1; F = F + c2; F = F + c3;
    }
    for (; k < count; k++) {
      const double *e = clist + (size_t)k * 4;
      F = F + mk(e[0], e[1], e[2]);
    }
    wave_lds_fence();
  }
  // attractorForceScaling (only if |F| > 1e-5, :319)
  if (norm(F) > 1e-5) {
    const bool cand = (best_i != 0x7fffffff) && (best_d == m);
    int bi;
    if (TILES == 1) {  // obstacle index == lane: lowest candidate lane wins
      const unsigned long long bm = __ballot(cand);
      bi = bm ? (__ffsll((long long)bm) - 1) : 0x7fffffff;
    } else {
      bi = wave_min64_i(cand ? best_i : 0x7fffffff);
    }
    if (bi == 0x7fffffff) {
      scale = 1;
    } else if (dot(g, v) <= 0.0 && nrm_v < C.vel_max - 0.1 * C.vel_max && dg > 0.15) {
      scale = 0.0;
    } else {
      const double w1 = 1 - portable_exp(-__builtin_sqrt(m) / C.shell);
      const int bl = bi & 63, bt = bi >> 6;
      V3 bp = mk(0.0, 0.0, 0.0);
#pragma unroll
      for (int t = 0; t < TILES; t++)
        if (t == bt) bp = mk(readlane_d(O.p[t].x, bl), readlane_d(O.p[t].y, bl), readlane_d(O.p[t].z, bl));
      const V3 ro = bp - p;
      double w2 = 1 - (dot(g, ro) / (dg * norm(ro)));
      w2 = w2 * w2;
      scale = w1 * w2;
    }
  }
}

}  // namespace pmaf

// pmaf_k_mw.hip -- k_rollout_mw<W, MATH, PLAIN>: W waves per agent (latency shape with MANY obstacles: BASELINE C3,
// 256 agents x 500 steps x 128 obstacles) and its launcher. Compiled per arithmetic policy (-DPMAF_MW_MATH=1|2|3,
// csrc/build.sh; 3 with -ffp-contract=fast); each object defines pmaf_k_launch_mw_m<policy>.
//
// Why: with 61..256 field obstacles the wave-per-agent kernel holds 2 or 4 obstacle slots per lane and its lone wave
// issues the per-obstacle instructions of every slot (C3: 769 instructions per step against the one-slot kernel's 447,
// profiles/r6_c3_strict_steploop.txt) while three quarters of the chip's 1024 SIMDs idle. Here an agent is a BLOCK of
// W = ceil(M / 64) waves on W SIMDs of one CU and every wave runs the ONE-slot step on its own <= 64 obstacles (<= 61:
// lanes 61..63 stay the tail's riders and the sweep's norms ride in the tail's sequence, pmaf_k_w64.hip; 62..64: the
// sweep takes its own). What an agent-step needs from ALL obstacles is exchanged ONCE per step through LDS with a
// single s_barrier:
//   pre-barrier  (wave-local) sweep, first-contact latches, circular-field terms compacted into the wave's OWN list
//                (ascending obstacle index), attractorForceScaling's weight for the wave's OWN closest obstacle, the
//                wave's record {min distance, weight, has-candidate, count};
//   barrier      s_waitcnt lgkmcnt(0) + s_barrier (path stores are NOT drained);
//   post-barrier (every wave, redundantly, on identical operands => identical bits in every wave) the weight of the
//                first wave that holds the agent's minimum, the ordered force sum over list 0, list 1, ... (= ascending
//                obstacle index, the reference's `force_ += curr_force` order), then the tail.
// Every wave therefore carries the full agent state and no second hand-off is needed. Records and lists are double
// buffered on the parity of the exchange count: a wave that is one exchange ahead writes the other buffer.
// What it buys and what it cannot (profiles/r4_ab_mw.txt): the step's critical path -- sweep, circular terms, ordered sum,
// tail: dependent chains -- is as long as before; only the ISSUE of the per-obstacle instructions is spread over W SIMDs.
// The one-wave two-slot kernel already hides most of its second slot in the first slot's dependency bubbles, so C3
// (128 obstacles) gains 5 %; 129..256 obstacles (four slots per lane) gain 26..32 %, 62..122 gain 11..13 %. The earlier
// two-wave experiments (NOTES: full split +14 %, helper wave +24 %) split the step by FUNCTION, which shortens no chain.
// Bit-exact with the oracle like every other kernel (tests/test_mw_gpu.py; tests/test_parity_gpu.py runs its
// many-obstacle cases through this kernel AND, with PMAF_MW=0, through the one-wave kernels, which stay for launches
// that cannot give every wave a SIMD of its own and for policy 0).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "pmaf_types.hpp"
#include "pmaf_device.hpp"
#include "pmaf_rollout_w64.hpp"

using namespace pmaf;

namespace {

constexpr int MW_REGION = 512;   // doubles per (wave, parity): 4 list chunks of 64 + the padding chunk + 192 scratch
constexpr int MW_SCRATCH = 320;  // first scratch double of a region
template <int W>
struct MwLds {                   // offsets in doubles from the block's dynamic LDS base
  static constexpr int REC = 0;                       // [2][W][4] closest-obstacle records
  static constexpr int CELL = REC + 2 * W * 4;        // [4] the waves' private minimum cells
  static constexpr int FIN = CELL + 4;                // [4] the waves' min_obs_dist_ at the end of the rollout
  static constexpr int LIST = FIN + 4;                // [W][2][MW_REGION]
  static constexpr int TAB = LIST + W * 2 * MW_REGION;  // Obstacle / GoalObstacle bodies: position mirror, closest table
};

#ifdef PMAF_MW_TIMERS   // timing experiments only: s_memtime around the hand-off (printed by agents 2 and 7)
struct MwTimers { unsigned long long t0 = 0, drain = 0, wait = 0, n = 0; };
#else
struct MwTimers {};
#endif

// all waves of the block; LDS traffic only (the path stores stay in flight)
__device__ __forceinline__ void mw_barrier(MwTimers &TM) {
#ifdef PMAF_MW_TIMERS
  unsigned long long a, m, b;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %1\n\ts_barrier\n\ts_memtime %2\n\ts_waitcnt lgkmcnt(0)"
               : "=s"(a), "=s"(m), "=s"(b) : : "memory");
  TM.drain += m - a; TM.wait += b - m; TM.n++;
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// circForce (B/src/cf_agent.cpp:72-108) + attractorForceScaling (:195-227) for one agent on W waves; see the file
// header. s_pre / ron_pre: |ro| and ro.normalized() of this lane's obstacle, computed by the previous step's tail.
// tab: position mirror of ALL field obstacles as of this step ([3][mpd]), cidx: this wave's view of the closest-other
// table or nullptr (Obstacle / GoalObstacle bodies only).
// PRE: |ro| and ro.normalized() come from the previous step's tail (<= 61 obstacles per wave, lanes 61..63 are the tail's
// riders); !PRE: all 64 lanes hold obstacles and the sweep takes its own norm (the riders still ride in the tail's sequence).
template <int W, int TYPE, int MATH, bool PRE>
__device__ __forceinline__ void circ_and_scale_mw(const int lane, const int w, const int base, const int left, const V3 p,
                                                  const V3 v, const double zv, const V3 g, const double dg, const V3 gn,
                                                  const PopConst &C, const double k_circ, const int n_obs, double *rot_g,
                                                  unsigned &known_bits, LaneObstacles<1> &O, double *lds, const int xp,
                                                  double &lane_min, V3 &F, double &scale, const ExpK &EK,
                                                  const double s_pre, const V3 ron_pre, const lmask gate_m,
                                                  const double *tab, const int mpd, const int32_t *cidx, MwTimers &TM) {
  typedef Mth<MATH> MT;
  typedef MwLds<W> L;
  constexpr int NONE = 0x7fffffff;   // (closest-other search)
  const int M = n_obs - 1;
  const int i = base + lane;
  double *mylist = lds + L::LIST + (w * 2 + xp) * MW_REGION;
  // ---- sweep geometry (circForce :76-88, attractorForceScaling :201-211), this wave's obstacles ----
  const lmask valid_m = gate_m & ((left >= 64) ? ~0ull : ((1ull << left) - 1ull));
  const V3 ro = O.p[0] - p;
  const V3 rv = v - O.v[0];
  double s = s_pre;
  V3 ron = ron_pre;
  // (62..64 obstacles per wave: the norms HERE, at the head of the step -- as a second sequence in the tail, next to the
  // riders', they lengthen the block the whole step waits for: C3 929.7 -> 961.0 us, measured)
  if (!PRE) MT::template norm_unit<true>(ro, s, ron);
  const lmask skip_m = PMAF_BAL(dot(ron, gn) < -0.01) & PMAF_BAL(dot(ro, rv) < -0.01);
  double d = s - (C.rad + O.r[0]);
  d = smax(d, 1e-5);
  const lmask closer_m = valid_m & PMAF_BAL(d < C.shell);
  const double best_d = PMAF_LANE(closer_m) ? d : C.shell;
  const double best_s = s, best_gr = dot(g, ro);
  const lmask live_m = valid_m & ~skip_m;
  if (PMAF_LANE(live_m & PMAF_BAL(d < lane_min))) lane_min = d;
  const lmask in_m = live_m & PMAF_BAL(d < C.shell);
  // the wave's minimum of the closest distances through its private LDS cell (pmaf_rollout_w64.hpp, LDSMIN): issued
  // here, read back behind the circular terms
  unsigned long long *min_cell = reinterpret_cast<unsigned long long *>(lds + L::CELL + w);
  wave_lds_fence();
  *reinterpret_cast<double *>(min_cell) = C.shell;
  __hip_atomic_fetch_min(min_cell, (unsigned long long)__double_as_longlong(best_d), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
  // ---- first contact: latch the rotation vector (:92-96, rare) ----
  const lmask need_latch_m = in_m & PMAF_BAL((known_bits & 1u) == 0u);
  if (PMAF_RARE(need_latch_m != 0ull)) {
    V3 cpos = O.p[0];
    if (TYPE == T_OBST || TYPE == T_GOALOBST) {
      // "nearest other field obstacle" (B/src/cf_agent.cpp:434-446 / :480-492) over ALL M obstacles: the other waves'
      // positions come out of the mirror (the bits their registers hold at this step)
      if (cidx) {
        const int c = cidx[(lane < left) ? i : base];
        cpos = mk(tab[c], tab[mpd + c], tab[2 * mpd + c]);
      } else {
        unsigned long long pend = need_latch_m;
        while (pend) {
          const int Lx = __ffsll((long long)pend) - 1;
          pend &= pend - 1;
          const int id = base + Lx;
          const V3 own = readlane_v3(O.p[0], Lx);
          double bd = 100.0;
          int bj = NONE;
          for (int j = lane; j < M; j += 64) {   // ascending per lane: the lowest index among a lane's exact minima stays
            const double dj = MT::norm(own - mk(tab[j], tab[mpd + j], tab[2 * mpd + j]));
            if (j != id && dj < bd) { bd = dj; bj = j; }
          }
          const double md = wave_min64(bd);
          const int mj = wave_min64_i((bj != NONE && bd == md) ? bj : NONE);
          const int c = (mj == NONE) ? 0 : mj;
          if (lane == Lx) cpos = mk(tab[c], tab[mpd + c], tab[2 * mpd + c]);
        }
      }
    }
    if (PMAF_LANE(need_latch_m)) {
      const V3 rot = calc_rot_vec_pre<MATH>(TYPE, p, n_obs, O.p[0], cpos, mk(O.qx[0], O.qy[0], O.qz[0]), ron, g, dg, gn);
      rot_g[i] = rot.x; rot_g[n_obs + i] = rot.y; rot_g[2 * n_obs + i] = rot.z;
      O.rx[0] = rot.x; O.ry[0] = rot.y; O.rz[0] = rot.z;
      known_bits |= 1u;
    }
  }
  // (the read-back of the wave minimum is issued HERE: its LDS round trip runs under the circular terms)
  wave_lds_fence();
  const double mw = __longlong_as_double((long long)__hip_atomic_load(min_cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  // attractorForceScaling's weight (:212-226) for THIS wave's closest obstacle, in front of the barrier: a second dependent
  // chain (sqrt, divide, exp, divide) next to the circular terms' in one block. The wave that holds the agent's closest
  // obstacle computes it from the agent's minimum distance (its own minimum is the agent's) and that obstacle's |ro|
  // and g.ro -- the reference's operands; the other waves' values are discarded behind the barrier.
  const lmask cand_m = closer_m & PMAF_BAL(best_d == mw);
  const int has_w = cand_m ? 1 : 0;
  double wgt_w;
  {
    const int bl = cand_m ? (__ffsll((long long)cand_m) - 1) : 63;
    const double sb = readlane_d(best_s, bl), gr = readlane_d(best_gr, bl);
    const double w1 = 1 - portable_exp_nonpos<MATH>(-MT::div_pos(MT::sqrt_pos(mw), C.shell), EK);
    double w2 = 1 - MT::div(gr, dg * sb);
    w2 = w2 * w2;
    wgt_w = w1 * w2;
  }
  // ---- circular-field terms (:97-106), every lane (lanes without a term store to their scratch entry) ----
  int count;
  {
    const V3 rot = mk(O.rx[0], O.ry[0], O.rz[0]);
    const double zrv = sqn(rv);
    double vn, rvn;
    MT::norm_rcp_zpos(zrv, vn, rvn);
    const V3 nv = MT::div3_n_pos(rv, vn, rvn);
    const V3 cur = current_vector<MATH, true>(TYPE, rv, g, ron, rot);
    const V3 c = MT::div_pos(k_circ, d * d) * unit_triple<MATH>(nv, cur);
    const lmask m = in_m & PMAF_BAL(zrv != 0);   // vel_norm != 0, :98
    const bool has_c = PMAF_LANE(m);
    const int sl = lane_rank(m);
    int li = ((sl >> 4) << 6) + (sl & 15);       // row-transposed list (pmaf_rollout_w64.hpp, DPPSUM)
    asm("" : "+v"(li));
    const int idx = has_c ? li : (MW_SCRATCH + (lane >> 4) * 48 + (lane & 15));
    PMAF_BOUND(sl < 64 && idx + 32 < MW_REGION);
    mylist[idx] = c.x; mylist[idx + 16] = c.y; mylist[idx + 32] = c.z;
    count = __popcll(m);
    // zero padding: the rest of the list's last chunk (the whole next chunk when the list ends on a chunk boundary,
    // an empty list's first chunk): the readers add chunk 0 of every list unconditionally
    if ((lane & 15) >= (count & 15)) mylist[((count >> 4) << 6) + lane] = 0.0;
  }
  // ---- this wave's closest-obstacle record ----
  {
    double *rec = lds + L::REC + (xp * W + w) * 4;
    rec[0] = mw; rec[1] = wgt_w; rec[2] = __hiloint2double(count, has_w);   // every lane, same address, same value
  }
  mw_barrier(TM);
  // ======== every wave: the agent's closest obstacle and its scaling value out of the records, the ordered sum ========
  double rm[W], rwg[W], rpk[W], e0[W], e1[W];
#pragma unroll
  for (int u = 0; u < W; u++) {
    const double *r = lds + L::REC + (xp * W + u) * 4;
    rm[u] = r[0]; rwg[u] = r[1]; rpk[u] = r[2];
  }
#pragma unroll
  for (int u = 0; u < W; u++) {
    const double *lu = lds + L::LIST + (u * 2 + xp) * MW_REGION;
    e0[u] = lu[lane]; e1[u] = lu[64 + lane];      // (chunk 1: used iff the list holds more than 16 terms)
  }
  // non-negative doubles order like their bit patterns (the in-wave minimum is the same unsigned-64 minimum)
  unsigned long long mb = (unsigned long long)__double_as_longlong(rm[0]);
#pragma unroll
  for (int u = 1; u < W; u++) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(rm[u]);
    mb = (b < mb) ? b : mb;
  }
  // lowest obstacle index among the exact minima = the first wave that holds one (waves own ascending index ranges);
  // that wave's weight was computed from the agent's minimum and its own candidate: the operands attractorForceScaling
  // has for the agent's closest obstacle
  bool any = false;
  double wgt = 0.0;
  int cnt[W];
#pragma unroll
  for (int u = W - 1; u >= 0; u--) {
    const bool take = ((unsigned long long)__double_as_longlong(rm[u]) == mb) && (__double2loint(rpk[u]) != 0);
    any = any || take; wgt = take ? rwg[u] : wgt;
    cnt[u] = __builtin_amdgcn_readfirstlane(__double2hiint(rpk[u]));
  }
  const bool stall = (dot(g, v) <= 0.0) && (zv < C.zv09_lt) && (dg > 0.15);
  const double sc = any ? (stall ? 0.0 : wgt) : 1.0;
  // F = ((0 + c_0) + c_1) + ... : one fused accumulate per entry (v_fmac_f64_dpp acc += row_newbcast:k(e) * 1.0, exact;
  // hazards as in pmaf_rollout_w64.hpp). The first list's first chunk as separate statements: the records' selects ride in
  // its second issue slots.
  double acc = 0.0;
  double one = 1.0;
  asm volatile("" : "+v"(one));
#define PMAF_FM1(K) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(e), "v"(one));
#define PMAF_FM(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
  // A chunk that nothing can be interleaved with (every chunk but the first list's first) leaves the accumulate's second
  // issue slot empty, so a scalar compare + branch behind every fourth entry rides for free while it is not taken, and
  // the chunk ends at the list's end rounded up to four entries (the padding is +0.0: exact) -- one taken branch against
  // up to twelve dependent accumulates. REM: entries of the list from this chunk's first one on (may exceed 16).
#define PMAF_MW_CHUNK(E, REM) asm volatile("s_nop 4\n\t" \
                     PMAF_FM(0) PMAF_FM(1) PMAF_FM(2) PMAF_FM(3) \
                     "s_cmp_lt_i32 %3, 5\n\ts_cbranch_scc1 .Lpmaf_mw_%=\n\t" \
                     PMAF_FM(4) PMAF_FM(5) PMAF_FM(6) PMAF_FM(7) \
                     "s_cmp_lt_i32 %3, 9\n\ts_cbranch_scc1 .Lpmaf_mw_%=\n\t" \
                     PMAF_FM(8) PMAF_FM(9) PMAF_FM(10) PMAF_FM(11) \
                     "s_cmp_lt_i32 %3, 13\n\ts_cbranch_scc1 .Lpmaf_mw_%=\n\t" \
                     PMAF_FM(12) PMAF_FM(13) PMAF_FM(14) PMAF_FM(15) \
                     ".Lpmaf_mw_%=:\n\ts_nop 1" \
                     : "+v"(acc) : "v"(E), "v"(one), "s"(REM) : "scc")
#pragma unroll
  for (int u = 0; u < W; u++) {
    if (u == 0) {
      double e = e0[0];
      asm volatile("s_nop 1" : "+v"(e));
      PMAF_FM1(0) PMAF_FM1(1) PMAF_FM1(2) PMAF_FM1(3) PMAF_FM1(4) PMAF_FM1(5) PMAF_FM1(6) PMAF_FM1(7)
      PMAF_FM1(8) PMAF_FM1(9) PMAF_FM1(10) PMAF_FM1(11) PMAF_FM1(12) PMAF_FM1(13) PMAF_FM1(14) PMAF_FM1(15)
    } else {
      PMAF_MW_CHUNK(e0[u], cnt[u]);
    }
    if (cnt[u] > 16) {
      const double *lu = lds + L::LIST + (u * 2 + xp) * MW_REGION;
      PMAF_MW_CHUNK(e1[u], cnt[u] - 16);
      for (int c16 = 32; PMAF_RARE(c16 < cnt[u]); c16 += 16) {
        const double e = lu[(c16 << 2) + lane];
        PMAF_MW_CHUNK(e, cnt[u] - c16);
      }
    }
  }
#undef PMAF_MW_CHUNK
#undef PMAF_FM
#undef PMAF_FM1
  asm volatile("s_nop 0" : "+v"(acc));
  F = mk(readlane_d(acc, 0), readlane_d(acc, 16), readlane_d(acc, 32));
  scale = (sqn(F) >= C.zf_gt) ? sc : scale;  // norm(F) > 1e-5
}

// the step loop: rollout_w64_body's one-slot (PRE) loop on W waves (pmaf_k_w64.hip has the commentary of every block)
template <int W, int TYPE, int MATH, int SENT, bool PLAIN, bool PRE>
__device__ __forceinline__ void rollout_mw_body(const DevView &D, const CostParams &CP, const int lane, const int w,
                                                const int pop, const int a, const int per) {
  extern __shared__ double smem[];
  typedef MwLds<W> L;
  typedef Mth<MATH> MT;
  const unsigned long long t_begin = wall_clock64();
  const int n_obs = D.n_obs;
  const int M = n_obs - 1;
  PopConst C = D.C;
  { double *f = reinterpret_cast<double *>(&C);
    for (int i = 0; i < (int)(sizeof(PopConst) / sizeof(double)); i++) asm volatile("" : "+v"(f[i])); }
  const ExpK EK = exp_consts_in_vgprs();
  const size_t pa = (size_t)pop * D.N + a;
  const double *src = D.obs_start + (size_t)pop * 7 * n_obs;
  const int32_t *ks = D.known_start + (size_t)pop * n_obs;
  double *rot_g = D.rot + pa * 3 * n_obs;
  const double *rnd_g = D.rnd + pa * 3 * n_obs;

  // this wave's obstacles: base .. base + left - 1, one per lane; PRE (per <= 61): lanes 61..63 are the tail's riders
  const int base = w * per;
  const int left = (M - base < per) ? ((M - base < 0) ? 0 : (M - base)) : per;
  const bool valid = lane < left;
  const int i = base + lane;
  const int ii = valid ? i : 0;
  LaneObstacles<1> O;
  O.p[0] = mk(src[ii], src[n_obs + ii], src[2 * n_obs + ii]);
  O.v[0] = mk(src[3 * n_obs + ii], src[4 * n_obs + ii], src[5 * n_obs + ii]);
  O.r[0] = src[6 * n_obs + ii];
  O.rx[0] = rot_g[ii]; O.ry[0] = rot_g[n_obs + ii]; O.rz[0] = rot_g[2 * n_obs + ii];
  if (TYPE == T_RANDOM) { O.qx[0] = rnd_g[ii]; O.qy[0] = rnd_g[n_obs + ii]; O.qz[0] = rnd_g[2 * n_obs + ii]; }
  else { O.qx[0] = 0.0; O.qy[0] = 0.0; O.qz[0] = 0.0; }
  unsigned known_bits = (valid && ks[ii]) ? 1u : 0u;
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

  double lane_min = C.shell;
  int n = 1;
  bool ran = false;
  if (w == 0 && lane == 0) { path[0] = p.x; path[1] = p.y; path[2] = p.z; }

  // any field obstacle of the POPULATION with a non-zero (or NaN) velocity component (every wave scans all of them)
  bool moving = false;
  for (int j = lane; j < M; j += 64)
    moving = moving || !(src[3 * n_obs + j] == 0.0 && src[4 * n_obs + j] == 0.0 && src[5 * n_obs + j] == 0.0);
  moving = wave_any(moving);

  // Obstacle / GoalObstacle bodies: the latch's scan needs every field obstacle's position as of the current step.
  // Mirror [2][3][mpd] in LDS: obstacles at rest -- buffer 0 the start positions (step 0), buffer 1 the positions after
  // one application of predictObstacles (p + (+-0) dt turns a -0.0 coordinate into +0.0 and is idempotent from then
  // on); moving obstacles -- buffer (step & 1), every wave writing its obstacles' NEXT positions in front of the step's
  // barrier (the exchange then runs on every step, gate or not, so that no wave is more than one step ahead).
  constexpr bool CLOSE = (TYPE == T_OBST || TYPE == T_GOALOBST);
  const int mpd = ((M + 63) & ~63) + 64;          // per component: M positions + one dump entry per lane
  double *ptab = smem + L::TAB;
  const int32_t *cidx = nullptr;
  if (CLOSE) {
    const int ti = valid ? i : (mpd - 64 + lane);
    const V3 p1 = O.p[0] + O.v[0] * C.dt;
    ptab[ti] = O.p[0].x; ptab[mpd + ti] = O.p[0].y; ptab[2 * mpd + ti] = O.p[0].z;
    ptab[3 * mpd + ti] = p1.x; ptab[4 * mpd + ti] = p1.y; ptab[5 * mpd + ti] = p1.z;
    if (!moving && D.closest_ok[pop] == 1) {     // closest-other table (k_manager): this wave's entries
      int32_t *s_cidx = reinterpret_cast<int32_t *>(ptab + 6 * mpd);
      if (valid) s_cidx[i] = D.closest_idx[(size_t)pop * n_obs + i];
      cidx = s_cidx;
    }
    __syncthreads();
  }

  V3 g = goal - p;
  double dg = MT::norm(g);
  double zv = sqn(v);
  double z_init = sqn(p - init_pos);
  V3 gn = (dg > 0.0) ? MT::div3(g, dg) : g;
  double s_pre = 0.0;
  V3 ron_pre = mk(0.0, 0.0, 0.0);
  double lane_scale = (lane == 61) ? (k_attr / k_damp) : 1.0;
  asm volatile("" : "+v"(lane_scale));
  if (PRE) {
    const double nz = (C.dt < 0.0) ? 0.0 : -0.0;
    if (lane == 63 || lane == 61) { O.p[0] = goal; O.v[0] = mk(nz, nz, nz); }
    MT::norm_unit(O.p[0] - p, s_pre, ron_pre);
  }
  V3 verr = attractor_velocity_error<MATH>(v, g, C, k_attr, k_damp);
  const double zsent_lt = D.zsent_lt[pop];
  const bool sent_reachable = (SENT == 1);
  V3 repel = mk(0.0, 0.0, 0.0);
  const double inv_shell = (SENT == 0) ? 0.0 : 1.0 / C.shell;
  if (sent_reachable) repel = sentinel_repel_m<MATH>(p, C, k_repel, sent_p, sent_r, zsent_lt, inv_shell);
  int xp = 0;   // parity of the number of exchanges done
  MwTimers TM;
#ifdef PMAF_MW_TIMERS
  TM.t0 = __builtin_amdgcn_s_memtime();
#endif
  while ((dg > 0.1) && (n < D.cap)) {
    const lmask gate_m = ~(PMAF_BAL(dg < C.approach) | (PMAF_BAL(zv < C.zvhalf_lt) & PMAF_BAL(z_init < C.zinit_lt)));
    V3 F = mk(0.0, 0.0, 0.0);
    double scale = 1.0;
    // gate closed (:315-317): no obstacle counts, nothing to exchange -- every wave decides that on the same state
    if ((gate_m != 0ull) || (CLOSE && moving)) {
      const double *tab = nullptr;
      if (CLOSE) {
        const int step = n - 1;
        if (moving) {   // the NEXT step's positions, in front of this step's barrier
          double *nx = ptab + ((step + 1) & 1) * 3 * mpd;
          const int ti = valid ? i : (mpd - 64 + lane);
          const V3 pn = O.p[0] + O.v[0] * C.dt;
          nx[ti] = pn.x; nx[mpd + ti] = pn.y; nx[2 * mpd + ti] = pn.z;
        }
        tab = ptab + (moving ? (step & 1) : (step > 0 ? 1 : 0)) * 3 * mpd;
      }
      circ_and_scale_mw<W, TYPE, MATH, PRE>(lane, w, base, left, p, v, zv, g, dg, gn, C, k_circ, n_obs, rot_g, known_bits, O,
                                       smem, xp, lane_min, F, scale, EK, s_pre, ron_pre, gate_m, tab, mpd, cidx, TM);
      xp ^= 1;
    }
    F = F + (mk(0.0, 0.0, 0.0) + repel);
    V3 acc;
    if (PLAIN) {
      F = F + (scale * k_damp) * verr;
      acc = F;
      const double az = sqn(F);
      if (PMAF_RARE(wave_any(az >= C.zacc_gt))) acc = acc * (13.0 / __builtin_sqrt(az));
    } else {
      const V3 Fa = F + (scale * k_damp) * verr;
      const bool attr = (k_attr != 0.0);
      F.x = attr ? Fa.x : F.x; F.y = attr ? Fa.y : F.y; F.z = attr ? Fa.z : F.z;
      acc = F;
      double az = sqn(F);
      if (PMAF_RARE(wave_any((C.mass != 1.0) || (az >= C.zacc_gt)))) {
        if (C.mass != 1.0) { acc = F / C.mass; az = sqn(acc); }
        if (az >= C.zacc_gt) acc = acc * (13.0 / __builtin_sqrt(az));
      }
    }
    const V3 half = ((0.5 * acc) * C.dt) * C.dt;
    const V3 new_pos = (p + half) + (v * C.dt);
    const V3 nv = v + acc * C.dt;
    p = new_pos;
    g = goal - p;
    O.p[0] = O.p[0] + O.v[0] * C.dt;   // predictObstacles, :270-276
    {
      const V3 vel_des = (k_attr / k_damp) * g;
      const bool l_nv = (lane == 62), l_des = (lane == 61);
      const V3 other = PRE ? (O.p[0] - p) : g;   // PRE: lanes 63 and 61 hold the goal, O.p - p = g there
      const V3 vec = l_nv ? nv : (other * lane_scale);
      V3 num = vec;
      num.x = (l_nv || l_des) ? C.vel_max : vec.x;
      const double zvec = sqn(vec);
      const double s = MT::sqrt(zvec);
      const double sd = PMAF_LANE(PMAF_BAL(zvec > 0.0) | (3ull << 61)) ? s : 1.0;
      const double rs = MT::rcp_for(sd);
      const V3 q = mk(MT::div_n(num.x, sd, rs), MT::div_n_pos(num.y, sd, rs), MT::div_n_pos(num.z, sd, rs));
      if (PRE) { s_pre = s; ron_pre = q; }
      const double vn = readlane_d(s, 62), f_nv = readlane_d(q.x, 62), f_des = readlane_d(q.x, 61);
      v = nv * ((vn > C.vel_max) ? f_nv : 1.0);
      dg = readlane_d(s, 63);
      gn = readlane_v3(q, 63);
      verr = vel_des * smin(1.0, f_des) - v;
    }
    zv = sqn(v);
    z_init = sqn(p - init_pos);
    PMAF_BOUND(n < D.cap);
    if (w == 0) { path[n * 3] = p.x; path[n * 3 + 1] = p.y; path[n * 3 + 2] = p.z; }   // wave-uniform: every lane, one address
    n++;
    ran = true;
    if (sent_reachable) {
      sent_p = sent_p + sent_v * C.dt;
      repel = sentinel_repel_m<MATH>(p, C, k_repel, sent_p, sent_r, zsent_lt, inv_shell);
    }
  }
#ifdef PMAF_MW_TIMERS
  if (lane == 0 && pop == 0 && (a == 7 || a == 2))
    printf("agent %d wave %d steps %d exchanges %llu | loop %llu ticks, LDS drain in front of the barrier %llu, barrier wait %llu\n", a, w, n - 1,
           TM.n, (unsigned long long)__builtin_amdgcn_s_memtime() - TM.t0, TM.drain, TM.wait);
#endif

  // min_obs_dist_ over the waves; known flags of this wave's obstacles
  const double wmin = wave_min64(lane_min);
  smem[L::FIN + w] = wmin;
  int32_t *ko = D.known_out + pa * n_obs;
  if (valid) ko[i] = (int32_t)(known_bits & 1u);
  __syncthreads();
  if (w != 0) return;
  double min_obs = smem[L::FIN];
#pragma unroll
  for (int u = 1; u < W; u++) min_obs = smin(min_obs, smem[L::FIN + u]);
  double cost_ws, path_len;
  path_cost_terms_w64<MATH>(lane, path, n, CP.ws, CP.k_workspace, smem + L::LIST, cost_ws, path_len);
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

}  // namespace

template <int W, int MATH, bool PLAIN, bool PRE>
__global__ __launch_bounds__(64 * W) void k_rollout_mw(DevView D, CostParams CP, int per) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pop = blockIdx.y;
  const int a = blockIdx.x;  // grid.x == N
  // the repulsive obstacle's reachability (per rollout, sentinel_reachable) picks a loop without any code for it
  const int n_obs = D.n_obs, M = n_obs - 1;
  const double *src = D.obs_start + (size_t)pop * 7 * n_obs;
  const V3 sp = mk(src[M], src[n_obs + M], src[2 * n_obs + M]);
  const V3 sv = mk(src[3 * n_obs + M], src[4 * n_obs + M], src[5 * n_obs + M]);
  const V3 p0 = mk(D.start_pos[pop * 3], D.start_pos[pop * 3 + 1], D.start_pos[pop * 3 + 2]);
  const PopConst C0 = D.C;
  const bool reach = sentinel_reachable(p0, sp, sv, D.zsent_lt[pop], C0, D.cap);
#define PMAF_BODY(T) \
  if (!reach) rollout_mw_body<W, T, MATH, 0, PLAIN, PRE>(D, CP, lane, w, pop, a, per); \
  else rollout_mw_body<W, T, MATH, 1, PLAIN, PRE>(D, CP, lane, w, pop, a, per)
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
}

#ifndef PMAF_MW_MATH
#error "compile with -DPMAF_MW_MATH=1|2|3"
#endif
#define PMAF_CAT2(a, b) a##b
#define PMAF_CAT(a, b) PMAF_CAT2(a, b)
#define PMAF_MW_LAUNCH PMAF_CAT(pmaf_k_launch_mw_m, PMAF_MW_MATH)
// waves: 2..4 waves per agent, per: field obstacles per wave (<= 64, waves * per >= M; <= 61: the riders' lanes are free and
// the sweep's norms ride in the tail's sequence, 62..64: the sweep takes its own)
bool PMAF_MW_LAUNCH(const DevView &D, const CostParams &cp, int waves, int per, bool plain, int lds_kb, hipStream_t s,
                    hipEvent_t e0, hipEvent_t e1) {
  const int M = D.n_obs - 1;
  if (waves < 2 || waves > 4 || per < 1 || per > 64 || waves * per < M) return false;
  const bool pre = per <= 61;
  const int mpd = ((M + 63) & ~63) + 64;
  const size_t lds = sizeof(double) * ((size_t)(2 * waves * 4 + 8) + (size_t)waves * 2 * MW_REGION + 6 * (size_t)mpd +
                                       (size_t)(mpd / 2 + 8));
  // Placement: ONE block per CU (the W waves on W of its 4 SIMDs). The LDS request enforces it whatever the dispatcher
  // would do by itself: 96 KB of the CU's 160 KB, so a second block does not fit. Round 4 let two two-wave blocks share
  // a CU (72 KB); round 5 measured that such a launch is 18 % slower than the one-wave two-slot kernel
  // (profiles/r5_mw_rule_sweep.txt) and the host no longer asks for it (pmaf_host.cpp, pick_mw).
  // `lds_kb`: the caller's override (0: this rule; timing experiments).
  size_t need = lds;
  { const size_t want = (size_t)(lds_kb > 0 ? lds_kb : 96) * 1024; if (want > need) need = want; }
  const dim3 grid((unsigned)D.N, (unsigned)D.P);
  // (the opt-in to more than 64 KB of dynamic LDS is per function AND device: remembered per device of the calling thread)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
#define PMAF_L1(WV, PL, PR) do { \
    static size_t lds_set[64] = {0}; \
    if (need > 64 * 1024 && need > lds_set[dev]) { \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_mw<WV, PMAF_MW_MATH, PL, PR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) != hipSuccess) return false; \
      lds_set[dev] = need; } \
    hipExtLaunchKernelGGL((k_rollout_mw<WV, PMAF_MW_MATH, PL, PR>), grid, dim3(64 * WV), (unsigned)need, s, e0, e1, 0, D, cp, per); } while (0)
#define PMAF_L(WV) do { if (plain) { if (pre) PMAF_L1(WV, true, true); else PMAF_L1(WV, true, false); } \
                        else { if (pre) PMAF_L1(WV, false, true); else PMAF_L1(WV, false, false); } } while (0)
  if (waves == 2) PMAF_L(2); else if (waves == 3) PMAF_L(3); else PMAF_L(4);
#undef PMAF_L
#undef PMAF_L1
  return true;
}

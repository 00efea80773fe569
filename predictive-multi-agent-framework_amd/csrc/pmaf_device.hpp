// pmaf_device.hpp -- gfx950 device functions of the circular-field agent step.
//
// One agent is evaluated by a group of LPA lanes of a wave64 (LPA = 1..64, a
// power of two): the lanes split the O(M) obstacle sweep of
// CfAgent::circForce / attractorForceScaling (B/src/cf_agent.cpp:72-108,
// 195-227; B/ = reference src/bimanual_planning_ros/), the O(1) terms
// (repelForce :159-181, attractorForce :183-193, updatePositionAndVelocity
// :253-268) are evaluated redundantly by every lane of the group so no
// broadcast is needed. Obstacles live in LDS as structure-of-arrays (one
// copy per wave: every agent's private obstacle copy in the reference evolves
// identically, cf_agent.cpp:270-276).
//
// Floating point: IEEE double, no FMA contraction (-ffp-contract=off), true
// divisions and square roots, and the reference's operation order (see
// oracle/pmaf_oracle.c header), so results are bit-identical to the CPU
// restatement; exp() (attractorForceScaling :220) is the portable_exp below (glibc's algorithm restated).
//
// The sequential `force_ += curr_force` of circForce (:106) is reproduced by
// adding the non-zero per-obstacle terms in ascending obstacle index
// (ballot + ordered lane walk) -- adding the zero terms of out-of-shell
// obstacles is an exact no-op.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmaf_types.hpp"

// -DPMAF_DEBUG_BOUNDS (tools/asan.sh; never in the product build): every index into a path, a list in LDS or a slot
// table is checked against its capacity and the wave traps (the launch fails with an exception the host reports) instead
// of writing out of bounds. The parity suite runs once through such a build per round (profiles/r6_asan.txt).
#ifdef PMAF_DEBUG_BOUNDS
#define PMAF_BOUND(c) do { if (!(c)) __builtin_trap(); } while (0)
#else
#define PMAF_BOUND(c) do { } while (0)
#endif

namespace pmaf {

struct V3 { double x, y, z; };

__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ double dot(V3 a, V3 b) {
#ifdef PMAF_DOT_RIGHT_ASSOC
  return a.x * b.x + (a.y * b.y + a.z * b.z);
#else
  return (a.x * b.x + a.y * b.y) + a.z * b.z;
#endif
}
__device__ __forceinline__ double sqn(V3 a) { return dot(a, a); }
__device__ __forceinline__ double norm(V3 a) { return __builtin_sqrt(sqn(a)); }
__device__ __forceinline__ V3 normalized(V3 a) {
  double z = sqn(a);
  if (z > 0.0) return a / __builtin_sqrt(z);
  return a;
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// v x (c x v), the circular term's direction (B/src/cf_agent.cpp:103). CONTRACTED policy only: for the unit vector v the
// triple product is c (v.v) - v (v.c) = c - v (v.c) up to the rounding of |v| = 1 -- one dot product and three fused
// multiply-adds instead of two cross products (18 -> 6 instructions per obstacle slot)
template <int MATH>
__device__ __forceinline__ V3 unit_triple(V3 v, V3 c);
// std::max(a,b) / std::min(a,b) semantics incl. NaN behaviour
__device__ __forceinline__ double smax(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double smin(double a, double b) { return (b < a) ? b : a; }

// ---- arithmetic policy ----------------------------------------------------
// How divisions and square roots are evaluated in the tuned rollout kernels.
//
// MATH_IEEE (0)  the compiler's expansions (v_div_scale .. v_div_fmas,
//                v_div_fixup; v_rsq + Goldschmidt with range scaling):
//                correctly rounded for every input. They serialise on VCC and
//                cost 65-72 / 110 cycles each on a lone wave even when several
//                are independent (tools/ubench.hip). PMAF_FLAG_IEEE_SEQUENCES.
// MATH_XACT (2)  DEFAULT. The same Newton / Goldschmidt iterations written out
//                without the range-scaling instructions, one refined reciprocal
//                shared by the three components of a vector divided by its
//                norm, v_div_fixup for zero / infinite / NaN operands. For
//                operands whose exponents lie within +-250 (every quantity of
//                this path for inputs in the validated range 2^-100..2^100:
//                lengths in metres, speeds, gains) the hardware's scaling is
//                the identity, so these sequences return the same bits as
//                MATH_IEEE -- the whole GPU parity suite is bit-exact in both
//                modes and test_xact_sequences_match_ieee sweeps the range.
//                Outside it (denormal-scale operands) results stay accurate to
//                rounding error but are not guaranteed bit-identical.
// MATH_FAST (1)  opt-in (PMAF_FLAG_FAST_MATH): v_rcp_f64 / v_rsq_f64 seeds
//                (2^-24) + two Newton iterations, no residual correction:
//                1-2 ulp per operation; tolerance parity only.
// MATH_FMA (3)   opt-in (PMAF_FLAG_CONTRACTED, round 4): MATH_FAST's sequences AND
//                the translation unit compiled with -ffp-contract=fast, so the
//                step's dot / cross / axpy forms retire as v_fma_f64 (dot 5 -> 3,
//                cross 9 -> 6, a + b * s 2 -> 1 instructions) -- the step is
//                issue-bound on instruction count and the FMA is the one
//                instruction that retires two flops. Where this policy is the
//                template argument the code may also drop bit-exactness-only
//                structure (the ordered force sum becomes a DPP tree). Tolerance
//                parity only (north star: selected trajectory <= 1e-5 m); the
//                real agent's step in k_manager stays strict.
// (enum MATH_IEEE / MATH_FAST / MATH_XACT / MATH_FMA: pmaf_types.hpp)

template <int MATH> struct Mth;
template <> struct Mth<MATH_IEEE> {
  static __device__ __forceinline__ double sqrt(double z) { return __builtin_sqrt(z); }
  static __device__ __forceinline__ double sqrt_pos(double z) { return __builtin_sqrt(z); }
  static __device__ __forceinline__ double div(double a, double b) { return a / b; }
  static __device__ __forceinline__ V3 div3(V3 a, double s) { return a / s; }
  static __device__ __forceinline__ double norm(V3 a) { return __builtin_sqrt(sqn(a)); }
  // s = |a| and the policy's helper value for dividing by s (unused here)
  static __device__ __forceinline__ void norm_rcp(V3 a, double &s, double &rs) { s = __builtin_sqrt(sqn(a)); rs = 0.0; }
  // as norm_rcp for a squared norm z the caller has at hand and tests itself (s is only divided by when z != 0)
  static __device__ __forceinline__ void norm_rcp_z(double z, double &s, double &rs) { s = __builtin_sqrt(z); rs = 0.0; }
  static __device__ __forceinline__ void norm_rcp_zpos(double z, double &s, double &rs) { s = __builtin_sqrt(z); rs = 0.0; }
  // the policy's helper value for dividing by s (see norm_rcp), for a divisor the caller has at hand
  static __device__ __forceinline__ double rcp_for(double) { return 0.0; }
  static __device__ __forceinline__ double div_n(double x, double s, double) { return x / s; }
  static __device__ __forceinline__ V3 div3_n(V3 a, double s, double) { return a / s; }
  // (the *_pos variants of the default policy are plain divisions here)
  static __device__ __forceinline__ double div_pos(double a, double b) { return a / b; }
  static __device__ __forceinline__ double div_n_pos(double x, double s, double) { return x / s; }
  static __device__ __forceinline__ V3 div3_n_pos(V3 a, double s, double) { return a / s; }
  // s = |a|, u = a.normalized()
  template <bool TP = false>
  static __device__ __forceinline__ void norm_unit(V3 a, double &s, V3 &u) {
    double z = sqn(a);
    s = __builtin_sqrt(z);
    u = (z > 0.0) ? (a / s) : a;
  }
  template <bool TP = false>
  static __device__ __forceinline__ V3 normalized(V3 a) { return pmaf::normalized(a); }
};
template <> struct Mth<MATH_XACT> {
  static __device__ __forceinline__ double sqrt(double z) {
    double y = __builtin_amdgcn_rsq(z);
    double g = z * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, z);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, z);
    g = __builtin_fma(d, h, g);
    return (z == 0.0 || z == __builtin_huge_val()) ? z : g;  // sqrt(+-0) = +-0, sqrt(inf) = inf
  }
  // sqrt(z) for a z that is known to be positive and finite, or whose root is discarded otherwise (the caller's
  // select says so): the same iteration without the zero / infinity select (3 instructions)
  static __device__ __forceinline__ double sqrt_pos(double z) {
    double y = __builtin_amdgcn_rsq(z);
    double g = z * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, z);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, z);
    g = __builtin_fma(d, h, g);
    return g;
  }
  static __device__ __forceinline__ double rcp_refined(double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
  }
  // a / b given r = rcp_refined(b); v_div_fixup supplies the IEEE results for
  // zero / infinite / NaN operands
  static __device__ __forceinline__ double div_r(double a, double b, double r) {
    double q = a * r;
    double e = __builtin_fma(-b, q, a);
    q = __builtin_fma(e, r, q);
    return __builtin_amdgcn_div_fixup(q, b, a);
  }
  static __device__ __forceinline__ double div(double a, double b) { return div_r(a, b, rcp_refined(b)); }
  // a / b WITHOUT the v_div_fixup, for the sites where it could only pass q through: b positive, finite and normal
  // (a norm behind a `squaredNorm > 0` select, a clamped squared distance, 2 - c in exp) or the quotient is discarded
  // whenever b is not (the selects say so at each call site), and a finite (a NaN stays a NaN). The residual is
  // formed as -(b q - a) instead of (a - b q) -- the same value, both negations are operand modifiers -- so that a
  // zero numerator keeps its sign through the refinement: (-0) / b = -0, as IEEE and the fixup give it
  // ((+0) + (-0) = +0 would turn it into +0). One instruction less per division on the step's dependent chains.
  static __device__ __forceinline__ double div_r_pos(double a, double b, double r) {
    double q = a * r;
    const double e = __builtin_fma(b, q, -a);
    return __builtin_fma(-e, r, q);
  }
  static __device__ __forceinline__ double div_pos(double a, double b) { return div_r_pos(a, b, rcp_refined(b)); }
  static __device__ __forceinline__ double div_n_pos(double x, double s, double rs) { return div_r_pos(x, s, rs); }
  static __device__ __forceinline__ V3 div3_n_pos(V3 a, double s, double rs) {
    return mk(div_r_pos(a.x, s, rs), div_r_pos(a.y, s, rs), div_r_pos(a.z, s, rs));
  }
  static __device__ __forceinline__ V3 div3(V3 a, double s) {
    const double r = rcp_refined(s);
    return mk(div_r(a.x, s, r), div_r(a.y, s, r), div_r(a.z, s, r));
  }
  static __device__ __forceinline__ double norm(V3 a) { return sqrt(sqn(a)); }
  // s = |a| and the refined reciprocal of s (shared by the divisions by this norm). It must be rcp_refined(s):
  // a reciprocal taken from the sqrt iteration (2h + one Newton step) passed 1.2e9 random a / sqrt(b) checks but is
  // NOT always the same double -- for s = 1 - 2^-53 (the norm of a cross product of two unit vectors!) both 1.0 and
  // 1 + 2^-52 are fixed points of the Newton step, it lands on the other one than v_rcp_f64 does, and the quotient
  // of a numerator on a rounding tie came out 1 ulp off (found by tools/fuzz_parity.py, now in the test suite).
  static __device__ __forceinline__ void norm_rcp(V3 a, double &s, double &rs) { s = sqrt(sqn(a)); rs = rcp_refined(s); }
  static __device__ __forceinline__ double rcp_for(double s) { return rcp_refined(s); }
  // (with sqrt_pos here too the one-slot kernels came out slower on one box -- C2 281.7 vs 279.1 us, C3 1276 vs 1250 us:
  // instruction scheduling, not arithmetic; measured per site with tools/ab.sh)
  static __device__ __forceinline__ void norm_rcp_z(double z, double &s, double &rs) { s = sqrt(z); rs = rcp_refined(s); }
  // the same for a caller that discards everything derived from s unless z is positive and finite (the circular term's
  // |rv|: the term only counts for squaredNorm != 0): no zero / infinity select behind the iteration (3 instructions)
  static __device__ __forceinline__ void norm_rcp_zpos(double z, double &s, double &rs) { s = sqrt_pos(z); rs = rcp_refined(s); }
  static __device__ __forceinline__ double div_n(double x, double s, double rs) { return div_r(x, s, rs); }
  static __device__ __forceinline__ V3 div3_n(V3 a, double s, double rs) {
    return mk(div_r(a.x, s, rs), div_r(a.y, s, rs), div_r(a.z, s, rs));
  }
  // normalized() returns the vector itself unless squaredNorm > 0. TP
  // (throughput shape, the group kernel): dividing by 1.0 instead does that
  // exactly (x / 1.0 == x for every x, zeros keep their sign, NaNs stay NaN)
  // with selects on the divisor and its reciprocal (4 instructions instead of
  // 6). The wave-per-agent kernel keeps the select behind the division, off
  // the sqrt -> divide chain its latency depends on.
  template <bool TP = false>
  static __device__ __forceinline__ void norm_unit(V3 a, double &s, V3 &u) {
    double z = sqn(a);
    s = sqrt(z);
    const double rs = TP ? 0.0 : rcp_refined(s);
    // (fixup-free divisions: the quotient is only used when squaredNorm > 0 -- then s is a positive normal and
    // |a_i| <= s -- or the divisor is the exact 1.0)
    if (TP) {
      // (round 3: ONE select, on the divisor; its refined reciprocal is computed behind it -- rcp_refined(1.0) is 1.0
      // exactly: v_rcp_f64 returns 1.0 or a neighbour, and the Newton steps round 1 - delta^2 to 1.0)
      const double sd = (z > 0.0) ? s : 1.0;
      u = div3_n_pos(a, sd, rcp_refined(sd));
    } else {
      V3 q = div3_n_pos(a, s, rs);
      u = (z > 0.0) ? q : a;
    }
  }
  // normalized() alone: the norm itself is not returned, so its zero / infinity select can go -- for squaredNorm == 0
  // the quotient is discarded (non-TP) or the divisor replaced by 1.0 (TP) before the garbage root is used
  template <bool TP = false>
  static __device__ __forceinline__ V3 normalized(V3 a) {
    const double z = sqn(a);
    const double s = sqrt_pos(z);
    if (TP) {
      const bool pos = z > 0.0;
      const double sd = pos ? s : 1.0;
      return div3_n_pos(a, sd, rcp_refined(sd));
    }
    const V3 q = div3_n_pos(a, s, rcp_refined(s));
    return (z > 0.0) ? q : a;
  }
};
template <> struct Mth<MATH_FAST> {
  static __device__ __forceinline__ double rcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return r;
  }
  // g ~ sqrt(z), y ~ 1/sqrt(z); z == 0 gives g = 0, y = 0
  static __device__ __forceinline__ void sqrt_rsqrt(double z, double &g, double &y) {
    double y0 = __builtin_amdgcn_rsq(z);
    y0 = (z > 0.0) ? y0 : 0.0;
    double gg = z * y0, h = 0.5 * y0;
    double r = __builtin_fma(-h, gg, 0.5);
    gg = __builtin_fma(gg, r, gg);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, gg, 0.5);
    gg = __builtin_fma(gg, r, gg);
    h = __builtin_fma(h, r, h);
    g = gg;
    y = h + h;
  }
  static __device__ __forceinline__ double sqrt(double z) { double g, y; sqrt_rsqrt(z, g, y); return g; }
  static __device__ __forceinline__ double sqrt_pos(double z) { return sqrt(z); }
  static __device__ __forceinline__ double div(double a, double b) { return a * rcp(b); }
  static __device__ __forceinline__ V3 div3(V3 a, double s) { double r = rcp(s); return a * r; }
  static __device__ __forceinline__ double norm(V3 a) { return sqrt(sqn(a)); }
  static __device__ __forceinline__ void norm_rcp(V3 a, double &s, double &rs) { sqrt_rsqrt(sqn(a), s, rs); }
  static __device__ __forceinline__ double rcp_for(double s) { return rcp(s); }
  static __device__ __forceinline__ void norm_rcp_z(double z, double &s, double &rs) { sqrt_rsqrt(z, s, rs); }
  static __device__ __forceinline__ void norm_rcp_zpos(double z, double &s, double &rs) { sqrt_rsqrt(z, s, rs); }
  static __device__ __forceinline__ double div_n(double x, double, double rs) { return x * rs; }
  static __device__ __forceinline__ V3 div3_n(V3 a, double, double rs) { return a * rs; }
  static __device__ __forceinline__ double div_pos(double a, double b) { return a * rcp(b); }
  static __device__ __forceinline__ double div_n_pos(double x, double, double rs) { return x * rs; }
  static __device__ __forceinline__ V3 div3_n_pos(V3 a, double, double rs) { return a * rs; }
  template <bool TP = false>
  static __device__ __forceinline__ void norm_unit(V3 a, double &s, V3 &u) {
    double y;
    sqrt_rsqrt(sqn(a), s, y);
    u = a * y;
  }
  template <bool TP = false>
  static __device__ __forceinline__ V3 normalized(V3 a) { double s; V3 u; norm_unit(a, s, u); return u; }
};

template <> struct Mth<MATH_FMA> : Mth<MATH_FAST> {};
template <int MATH>
__device__ __forceinline__ V3 unit_triple(V3 v, V3 c) {
  if constexpr (MATH == MATH_FMA) {
    const double vc = dot(v, c);
    return mk(c.x - v.x * vc, c.y - v.y * vc, c.z - v.z * vc);
  } else {
    return cross(v, cross(c, v));
  }
}

// exp() of attractorForceScaling (B/src/cf_agent.cpp:220). The reference calls std::exp, i.e. its platform's libm. Rounds
// 1-4 evaluated an exp of their own (0.81 ulp) and accepted a last-bit difference against any libm on 5-10 % of the
// arguments; round 5 RESTATES the libm the reference runs on instead: glibc >= 2.28's exp (sysdeps/ieee754/dbl-64/e_exp.c,
// the variant compiled with FMA contraction that the ifunc selects on x86-64 CPUs with FMA), operation for operation --
//   kd = fma(x, N / ln2, 0x1.8p52); ki = bits(kd); kd -= 0x1.8p52;  r = fma(kd, -ln2lo / N, fma(kd, -ln2hi / N, x));
//   (tail, sbits) = T[ki % N], sbits += ki << 45;  tmp = fma(r^4, C4 + r C5, fma(C2 + r C3, r^2, tail + r));
//   exp = fma(scale, tmp, scale)                    (N = 128; data: pmaf_exp_table.hpp, tools/gen_exp_table.py)
// -- only correctly rounded IEEE operations and integer bit operations, the same function as
// oracle/pmaf_oracle.c:pmaf_portable_exp, so host and gfx950 produce identical bits, AND the bits of the host libm's exp()
// wherever that libm is this algorithm (checked on 2e8 arguments on the build image, glibc 2.35): there the oracle's
// libm mode and its portable mode are one function and the kernels are bit-exact against the reference-faithful mode.
// 0.511 ulp. The dependent chain is shorter than the degree-11 Horner form's (kd -> r -> r^2 -> two fused stages -> the
// result: 8 operations deep instead of 16) at about the same instruction count; the table entry is a scalar load in the
// wave-per-agent kernels (the argument is wave-uniform) and an LDS read in the lane-group kernels.
// Range: x is clamped to >= -500 (below -512 glibc takes a subnormal-safe path with an unfused last step; the only caller
// forms 1 - exp(x), which is 1.0 for every x < -37.4, so the clamp changes no result and keeps one straight-line
// sequence); above 709.78: +inf; NaN propagates (non-NONPOS form).
#include "pmaf_exp_table.hpp"
struct ExpK {
  double invln2N, shift, neghi, neglo, c2, c3, c4, c5;
};
__device__ __forceinline__ ExpK exp_consts() {
  ExpK K;
  K.invln2N = 0x1.71547652b82fep+7; K.shift = 0x1.8p52; K.neghi = -0x1.62e42fefa0000p-8; K.neglo = -0x1.cf79abc9e3b3ap-47;
  K.c2 = 0x1.ffffffffffdbdp-2; K.c3 = 0x1.555555555543cp-3; K.c4 = 0x1.55555cf172b91p-5; K.c5 = 0x1.1111167a4d017p-7;
  return K;
}
// the constants pinned in VGPRs across a kernel's step loop: as literals they are scalar values the compiler keeps in --
// and spills from -- the SGPR file
__device__ __forceinline__ ExpK exp_consts_in_vgprs() {
  ExpK K = exp_consts();
  asm volatile("" : "+v"(K.invln2N), "+v"(K.shift), "+v"(K.neghi), "+v"(K.neglo), "+v"(K.c2), "+v"(K.c3), "+v"(K.c4), "+v"(K.c5));
  return K;
}
// the same data as an LDS table (kernels that have no registers to hold eight more loop-invariant doubles -- the group
// kernel sits at its two-waves-per-SIMD VGPR budget -- and every kernel whose exp argument differs from lane to lane):
// [0..7] the constants, [8 + 2k], [9 + 2k] the table; the constants are fetched right where the chain uses them
// (exp_consts_from_lds; the address passes through an empty asm so that the reads stay in the step loop)
constexpr int EXPK_N = PMAF_EXP_DATA_WORDS;
__device__ __forceinline__ void exp_consts_to_lds(double *tab, int lane) {
  for (int i = lane; i < EXPK_N; i += 64) tab[i] = __longlong_as_double((long long)PMAF_EXP_DATA[i]);
}
struct ExpKLds {
  unsigned off;    // LDS byte address of the table
  __device__ __forceinline__ void tie(double &p) { asm volatile("" : "+v"(off), "+v"(p)); }
  __device__ __forceinline__ double get(int i) const {
    typedef const __attribute__((address_space(3))) double *lds_cptr;
    return reinterpret_cast<lds_cptr>(static_cast<uintptr_t>(off))[i];
  }
  __device__ __forceinline__ void entry(unsigned idx, double &tail, unsigned long long &sb) const {   // one ds_read_b128
    typedef const __attribute__((address_space(3))) double *lds_cptr;
    lds_cptr e = reinterpret_cast<lds_cptr>(static_cast<uintptr_t>(off)) + 8 + 2 * idx;
    tail = e[0];
    sb = (unsigned long long)__double_as_longlong(e[1]);
  }
};
__device__ __forceinline__ ExpKLds exp_consts_from_lds(const double *tab) {
  typedef const __attribute__((address_space(3))) double *lds_cptr;
  ExpKLds K;
  K.off = (unsigned)reinterpret_cast<uintptr_t>((lds_cptr)tab);
  return K;
}
// the constants as values (literals / SGPRs, or VGPRs with exp_consts_in_vgprs). UNIFORM: the caller's argument is the
// same in every lane (one agent per wave): the table entry is fetched by ONE scalar load (s_load_dwordx4 out of the
// constant data) instead of a vector load per lane
template <bool UNIFORM>
struct ExpKRegs {
  const ExpK K;
  __device__ __forceinline__ void tie(double &) {}
  __device__ __forceinline__ double get(int i) const {
    switch (i) {
      case 0: return K.invln2N; case 1: return K.shift; case 2: return K.neghi; case 3: return K.neglo; case 4: return K.c2;
      case 5: return K.c3; case 6: return K.c4; default: return K.c5;
    }
  }
  __device__ __forceinline__ void entry(unsigned idx, double &tail, unsigned long long &sb) const {
    const unsigned i = UNIFORM ? (unsigned)__builtin_amdgcn_readfirstlane((int)idx) : idx;
    tail = __longlong_as_double((long long)PMAF_EXP_DATA[8 + 2 * i]);
    sb = PMAF_EXP_DATA[9 + 2 * i];
  }
};
// NONPOS: the caller guarantees x <= 0 and not NaN, or discards the result otherwise (attractorForceScaling's
// -sqrt(d) / shell with d in [1e-5, shell)): the overflow select and the NaN replacement drop out
template <int MATH, class KT, bool NONPOS = false>
__device__ __forceinline__ double portable_exp_k(double x, KT K) {
  double xs0 = x;
  K.tie(xs0);
  x = xs0;
  const double xlo = __builtin_fmax(x, -500.0);                          // v_max_f64 (a NaN is replaced: see below)
  const double xs = NONPOS ? xlo : __builtin_fmin(xlo, 0x1.62e42fefa39efp+9);   // exp's overflow threshold (the scale's exponent is handled below: x >= 512)
  double kd = __builtin_fma(xs, K.get(0), K.get(1));
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd = kd - K.get(1);
  double tail;
  unsigned long long sb;
  K.entry((unsigned)ki & (PMAF_EXP_N - 1), tail, sb);
  double r = __builtin_fma(kd, K.get(2), xs);
  r = __builtin_fma(kd, K.get(3), r);
  K.tie(r);
  const unsigned long long sbits = sb + (ki << 45);
  const double a = __builtin_fma(K.get(5), r, K.get(4));                 // C2 + r C3
  const double tr = r + tail;
  const double r2 = r * r;
  const double b = __builtin_fma(r, K.get(7), K.get(6));                 // C4 + r C5
  const double c = __builtin_fma(a, r2, tr);
  const double r4 = r2 * r2;
  const double tmp = __builtin_fma(r4, b, c);
  if (NONPOS) return __builtin_fma(__longlong_as_double((long long)sbits), tmp, __longlong_as_double((long long)sbits));
  // x >= 512: glibc's specialcase(), k > 0 branch -- 2^(k/N)'s exponent field would overflow just below the threshold
  // (k / N = 1024: `sbits` = +inf's pattern, fma(inf, tmp < 0, inf) = NaN), so the scale is formed 2^-1009 lower
  // and the result multiplied back up; same operations as oracle/pmaf_oracle.c:pmaf_portable_exp. Selects, no branch.
  const bool top = xs >= 512.0;
  const unsigned long long sb2 = top ? sbits - (1009ull << 52) : sbits;
  const double scale = __longlong_as_double((long long)sb2);
  const double res0 = __builtin_fma(scale, tmp, scale);
  const double res = top ? 0x1p1009 * res0 : res0;
  // above exp's overflow threshold: +inf; NaN in, NaN out (the clamps return their other operand for a NaN)
  const double big = (x > 0x1.62e42fefa39efp+9) ? __builtin_inf() : res;
  return (x != x) ? x : big;
}
// portable_exp_nonpos in two halves (the wave-per-agent step): `begin` forms k, requests the table entry and evaluates
// everything that does not need it; the caller puts independent work (the scaling's second factor: a division sequence)
// between the halves, `end` consumes the entry -- the scalar load's latency is covered instead of waited out. Same
// operations in the same order as portable_exp_k: same bits.
struct ExpPend { double r, a, b, r2, r4, tail; unsigned long long sbits; };
template <class KT>
__device__ __forceinline__ ExpPend portable_exp_nonpos_begin(double x, KT K) {
  ExpPend E;
  const double xs = __builtin_fmax(x, -500.0);
  double kd = __builtin_fma(xs, K.get(0), K.get(1));
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  unsigned long long sb;
  K.entry((unsigned)ki & (PMAF_EXP_N - 1), E.tail, sb);
  kd = kd - K.get(1);
  double r = __builtin_fma(kd, K.get(2), xs);
  r = __builtin_fma(kd, K.get(3), r);
  E.r = r;
  E.sbits = sb + (ki << 45);
  E.a = __builtin_fma(K.get(5), r, K.get(4));
  E.r2 = r * r;
  E.b = __builtin_fma(r, K.get(7), K.get(6));
  E.r4 = E.r2 * E.r2;
  return E;
}
__device__ __forceinline__ double portable_exp_nonpos_end(const ExpPend &E) {
  const double tr = E.r + E.tail;
  const double c = __builtin_fma(E.a, E.r2, tr);
  const double tmp = __builtin_fma(E.r4, E.b, c);
  const double scale = __longlong_as_double((long long)E.sbits);
  return __builtin_fma(scale, tmp, scale);
}
__device__ __forceinline__ ExpPend portable_exp_nonpos_begin(double x, const ExpK &K) { return portable_exp_nonpos_begin<ExpKRegs<true>>(x, ExpKRegs<true>{K}); }
__device__ __forceinline__ ExpPend portable_exp_nonpos_begin(double x, const ExpKLds &K) { return portable_exp_nonpos_begin<ExpKLds>(x, K); }
template <int MATH = MATH_IEEE>
__device__ __forceinline__ double portable_exp(double x, const ExpK &K) { return portable_exp_k<MATH, ExpKRegs<true>>(x, ExpKRegs<true>{K}); }
template <int MATH = MATH_IEEE>
__device__ __forceinline__ double portable_exp(double x, const ExpKLds &K) { return portable_exp_k<MATH, ExpKLds>(x, K); }
template <int MATH = MATH_IEEE>
__device__ __forceinline__ double portable_exp_nonpos(double x, const ExpK &K) { return portable_exp_k<MATH, ExpKRegs<true>, true>(x, ExpKRegs<true>{K}); }
template <int MATH = MATH_IEEE>
__device__ __forceinline__ double portable_exp_nonpos(double x, const ExpKLds &K) { return portable_exp_k<MATH, ExpKLds, true>(x, K); }
// (no constants handed in: the generic kernels and the self-test -- the argument may differ from lane to lane)
template <int MATH = MATH_IEEE>
__device__ __forceinline__ double portable_exp(double x) { return portable_exp_k<MATH, ExpKRegs<false>>(x, ExpKRegs<false>{exp_consts()}); }

// portable_exp_nonpos with its data requested in STAGES (the multi-slot wave-per-agent kernels, whose constants come out
// of LDS at the point of use): the reduction's four constants before the caller forms the argument (exp_head, tied to the
// argument's INPUT), the polynomial's four as soon as the argument exists, the table entry as soon as k does -- a lone
// wave does not sit out an LDS round trip (~20 issue slots) in front of each batch. Same operations in the same order as
// portable_exp_k: same bits.
struct ExpHead { double invln2N, shift, neghi, neglo; };
__device__ __forceinline__ ExpHead exp_head(ExpKLds &K, double &input) {
  K.tie(input);
  ExpHead H; H.invln2N = K.get(0); H.shift = K.get(1); H.neghi = K.get(2); H.neglo = K.get(3);
  return H;
}
__device__ __forceinline__ ExpHead exp_head(const ExpK &K, double &) {
  ExpHead H; H.invln2N = K.invln2N; H.shift = K.shift; H.neghi = K.neghi; H.neglo = K.neglo;
  return H;
}
template <class KT>
__device__ __forceinline__ double portable_exp_nonpos_staged(double x, KT K, const ExpHead H) {
  K.tie(x);
  const double c2 = K.get(4), c3 = K.get(5), c4 = K.get(6), c5 = K.get(7);
  const double xs = __builtin_fmax(x, -500.0);
  double kd = __builtin_fma(xs, H.invln2N, H.shift);
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  double tail;
  unsigned long long sb;
  K.entry((unsigned)ki & (PMAF_EXP_N - 1), tail, sb);
  kd = kd - H.shift;
  double r = __builtin_fma(kd, H.neghi, xs);
  r = __builtin_fma(kd, H.neglo, r);
  const unsigned long long sbits = sb + (ki << 45);
  const double a = __builtin_fma(c3, r, c2);
  const double tr = r + tail;
  const double r2 = r * r;
  const double b = __builtin_fma(r, c5, c4);
  const double c = __builtin_fma(a, r2, tr);
  const double r4 = r2 * r2;
  const double tmp = __builtin_fma(r4, b, c);
  const double scale = __longlong_as_double((long long)sbits);
  return __builtin_fma(scale, tmp, scale);
}
__device__ __forceinline__ double portable_exp_nonpos_staged(double x, const ExpK &K, const ExpHead H) {
  return portable_exp_nonpos_staged<ExpKRegs<true>>(x, ExpKRegs<true>{K}, H);
}
__device__ __forceinline__ double portable_exp_nonpos_staged(double x, const ExpKLds &K, const ExpHead H) {
  return portable_exp_nonpos_staged<ExpKLds>(x, K, H);
}

enum : int { T_REAL = 0, T_GOAL = 1, T_OBST = 2, T_GOALOBST = 3, T_VEL = 4, T_RANDOM = 5, T_HAD = 6 };

// population-wide scalars (CfManager::init arguments)
// (struct PopConst: pmaf_types.hpp)

// LDS-resident obstacle table, structure of arrays, n_obs entries each
struct ObsTab {
  double *px, *py, *pz, *vx, *vy, *vz, *r;
  __device__ __forceinline__ V3 pos(int i) const { return mk(px[i], py[i], pz[i]); }
  __device__ __forceinline__ V3 vel(int i) const { return mk(vx[i], vy[i], vz[i]); }
};

__device__ __forceinline__ ObsTab carve_obstab(double *base, int n_obs) {
  ObsTab t;
  t.px = base; t.py = base + n_obs; t.pz = base + 2 * n_obs;
  t.vx = base + 3 * n_obs; t.vy = base + 4 * n_obs; t.vz = base + 5 * n_obs;
  t.r = base + 6 * n_obs;
  return t;
}

// ---- cross-lane helpers -------------------------------------------------
// wave votes straight on the predicate (round 3): HIP's __any / __ballot take an int, and the bool -> int -> "!= 0"
// round trip stays in the code as v_cndmask + v_cmp_ne_u32 per vote (seen in the step loop's disassembly); for a lone
// wave every instruction is an issue slot
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

template <int LPA>
__device__ __forceinline__ double group_min(double v) {
#pragma unroll
  for (int off = LPA / 2; off > 0; off >>= 1) {
    double o = __shfl_xor(v, off);
    v = (o < v) ? o : v;
  }
  return v;
}

// (value, index) argmin with "first index wins" among equal values
template <int LPA>
__device__ __forceinline__ void group_argmin(double &d, int &idx) {
#pragma unroll
  for (int off = LPA / 2; off > 0; off >>= 1) {
    double od = __shfl_xor(d, off);
    int oi = __shfl_xor(idx, off);
    bool take = (od < d) || (od == d && oi < idx);
    d = take ? od : d;
    idx = take ? oi : idx;
  }
}

// F += c over the lanes of each group that have has_c set, in ascending lane
// (= ascending obstacle index within the tile) order.
template <int LPA>
__device__ __forceinline__ void ordered_accumulate(V3 &F, V3 c, bool has_c, int grp) {
  unsigned long long m = __ballot(has_c);
  if (LPA == 64) {
    while (m) {
      int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      F.x = F.x + readlane_d(c.x, src);
      F.y = F.y + readlane_d(c.y, src);
      F.z = F.z + readlane_d(c.z, src);
    }
  } else if (LPA == 1) {
    if (has_c) F = F + c;
  } else {
    const unsigned long long gm = (LPA >= 64) ? ~0ull : ((1ull << (LPA & 63)) - 1ull);
    unsigned long long sub = (m >> (grp * LPA)) & gm;
    while (__any(sub != 0ull)) {
      int src = grp * LPA + (sub ? (__ffsll((long long)sub) - 1) : 0);
      double cx = __shfl(c.x, src), cy = __shfl(c.y, src), cz = __shfl(c.z, src);
      if (sub) {
        F.x = F.x + cx; F.y = F.y + cy; F.z = F.z + cz;
      }
      sub &= sub - 1ull;
    }
  }
}

// ---- DPP wave reductions (wave64, no LDS crossbar round trips) -------------
// dpp_ctrl encodings: quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror
// 0x140, row_bcast15 0x142, row_bcast31 0x143 (GFX9 / CDNA).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
// Reductions run on 32-bit unsigned keys with the DPP operand fused into the
// min (v_min_u32_dpp: ONE VALU instruction per stage; an FP64 min needs two
// v_mov_dpp, two canonicalising v_max and the v_min per stage, measured 49
// cycles per stage on a lone wave). Non-negative, non-NaN doubles order like
// their bit patterns, so min(double) = min over the high words, then min over
// the low words of the lanes that hold the minimal high word.
// `old` = UINT_MAX (the identity of umin) lets the compiler's DPP combiner fold
// the v_mov_dpp into the v_min_u32 -- and, unlike inline asm, keeps the
// instructions visible to its hazard recogniser (DPP after a VALU write needs
// 2 wait states; a lane select read from an SGPR that a v_readlane just wrote
// needs 4: an inline-asm version of this reduction produced wrong lanes under
// some schedules).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_u32(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, CTRL, ROW_MASK, 0xf, false);
  return o < v ? o : v;
}

// minimum over the 64 lanes, returned wave-uniform
__device__ __forceinline__ unsigned wave_min64_u32(unsigned v) {
  v = dpp_min_u32<0xB1, 0xf>(v);   // lane ^ 1 within quads
  v = dpp_min_u32<0x4E, 0xf>(v);   // lane ^ 2 within quads
  v = dpp_min_u32<0x141, 0xf>(v);  // row_half_mirror: other quad of the 8-lane half
  v = dpp_min_u32<0x140, 0xf>(v);  // row_mirror: other half of the 16-lane row
  v = dpp_min_u32<0x142, 0xa>(v);  // row_bcast15 -> rows 1,3
  v = dpp_min_u32<0x143, 0xc>(v);  // row_bcast31 -> rows 2,3; lane 63 holds the total
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// v >= +0.0 and not NaN in every lane
__device__ __forceinline__ double wave_min64(double v) {
  const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
  const unsigned H = wave_min64_u32(hi);
  const unsigned L = wave_min64_u32((hi == H) ? lo : 0xffffffffu);
  return __hiloint2double((int)H, (int)L);
}
// v >= 0 in every lane
__device__ __forceinline__ int wave_min64_i(int v) { return (int)wave_min64_u32((unsigned)v); }

// ---- heuristics ----------------------------------------------------------
// currentVector, B/src/cf_agent.cpp:389-406 (Goal), 414-426 (Obstacle),
// 463-475 (GoalObstacle), 520-537 (Vel), 545-557 (Random), 585-597 (Had).
// to_obs = normalized(obstacle - agent_pos), identical to the value the
// reference recomputes inside each currentVector.
template <int MATH = MATH_IEEE, bool TP = false>
__device__ __forceinline__ V3 current_vector(int type, V3 agent_vel, V3 goal_vec, V3 to_obs, V3 rot) {
  typedef Mth<MATH> M;
  if (type == T_GOAL || type == T_VEL) {
    V3 cur;
    if (type == T_GOAL) {
      cur = goal_vec - to_obs * dot(to_obs, goal_vec);
    } else {
      V3 nvel = M::template normalized<TP>(agent_vel);
      cur = nvel - to_obs * dot(nvel, to_obs);
    }
    // `if (cur.norm() < 1e-10) cur = (0,0,1); return cur.normalized()` with ONE
    // square root: (0,0,1).normalized() is (0,0,1) exactly, and otherwise
    // normalized() divides by the same sqrt(squaredNorm(cur)) the test compared
    if constexpr (TP && MATH == MATH_XACT) {
      // tuned kernels (round 3): the test on the SQUARED norm -- sqrt is correctly rounded and monotone, so
      // (sqrt(z) < 1e-10) == (z < Z10) with Z10 the smallest double whose root is >= 1e-10 (0x1.79ca10c924223p-67, the
      // double 1e-20; host check: tests/test_oracle_properties.py) -- which takes the zero / infinity select out of
      // the root, and the `squaredNorm > 0` select out of normalized(): z == 0 is below the threshold, and for a NaN z
      // the quotient is NaN in every component, as is the vector normalized() would return (to_obs * NaN is NaN in
      // every component, so cur is)
      const double z = sqn(cur);
      const double s = M::sqrt_pos(z);
      const V3 q = M::div3_n_pos(cur, s, M::rcp_refined(s));
      return (z < 0x1.79ca10c924223p-67) ? mk(0.0, 0.0, 1.0) : q;
    }
    double s;
    V3 u;
    M::template norm_unit<TP>(cur, s, u);
    return (s < 1e-10) ? mk(0.0, 0.0, 1.0) : u;
  } else if (type == T_OBST || type == T_GOALOBST || type == T_RANDOM || type == T_HAD) {
    return M::template normalized<TP>(cross(to_obs, rot));
  }
  return mk(0.0, 0.0, 0.0);
}

// nearest other field obstacle by centre distance, cf_agent.cpp:434-446 / :480-492
__device__ __forceinline__ int closest_other(const ObsTab &T, int n_obs, int id, V3 oid) {
  double min_dist = 100.0;
  int closest = 0;
  for (int i = 0; i < n_obs - 1; i++) {
    if (i != id) {
      double d = norm(oid - T.pos(i));
      if (min_dist > d) { min_dist = d; closest = i; }
    }
  }
  return closest;
}

// calculateRotationVector, B/src/cf_agent.cpp:408-412 (Goal), 428-461
// (Obstacle), 477-518 (GoalObstacle), 539-543 (Vel), 559-566 (Random),
// 599-611 (Had)
// own_pos = position of obstacle `id`; closest_pos = position of the field
// obstacle nearest to it (closest_other; only read by the Obstacle and
// GoalObstacle heuristics). The tuned kernels find it with a cooperative
// search over the lanes' register copies, the generic kernel / k_manager scan
// the LDS table (calc_rot_vec below).
template <int MATH = MATH_IEEE>
__device__ __forceinline__ V3 calc_rot_vec_c(int type, V3 agent_pos, V3 goal_pos, int n_obs, V3 own_pos,
                                             V3 closest_pos, V3 rand_vec) {
  typedef Mth<MATH> M;
  if (type == T_GOAL || type == T_VEL) return mk(0.0, 0.0, 1.0);
  if (type == T_OBST) {
    if (n_obs < 2) return mk(0.0, 0.0, 1.0);
    V3 obstacle_vec = closest_pos - own_pos;
    V3 to_obs = M::normalized(own_pos - agent_pos);
    V3 cur = to_obs * dot(obstacle_vec, to_obs) - obstacle_vec;
    return M::normalized(cross(cur, to_obs));
  }
  if (type == T_GOALOBST) {
    V3 obstacle_vec = closest_pos - own_pos;
    V3 to_obs = M::normalized(own_pos - agent_pos);
    V3 obst_cur = to_obs * dot(obstacle_vec, to_obs) - obstacle_vec;
    V3 goal_vec = goal_pos - agent_pos;
    V3 goal_cur = goal_vec - to_obs * dot(to_obs, goal_vec);
    V3 cur = M::normalized(goal_cur) + M::normalized(obst_cur);
    // `if (cur.norm() < 1e-10) cur = (0,0,1); cur.normalize()` with one square root (see current_vector)
    double s;
    V3 u;
    M::norm_unit(cur, s, u);
    cur = (s < 1e-10) ? mk(0.0, 0.0, 1.0) : u;
    return M::normalized(cross(cur, to_obs));
  }
  if (type == T_RANDOM) {
    V3 goal_vec = M::normalized(goal_pos - agent_pos);
    return cross(goal_vec, rand_vec);
  }
  if (type == T_HAD) {
    V3 obs_pos = own_pos;
    V3 goal_vec = goal_pos - agent_pos;
    V3 rob_obs = obs_pos - agent_pos;
    double gn = M::norm(goal_vec);
    V3 d = (agent_pos + goal_vec * M::div(dot(rob_obs, goal_vec), gn * gn)) - obs_pos;
    V3 c = cross(d, goal_vec);
    return M::div3(c, M::norm(c));
  }
  return mk(0.0, 0.0, 0.0);
}

// calc_rot_vec_c for callers that already hold the step's geometry (the tuned kernels' first-contact latch): to_obs =
// normalized(own_pos - agent_pos) (the sweep's ro.normalized() of this slot), goal_vec = goal_pos - agent_pos, goal_dist
// = norm(goal_vec), goal_dir = goal_vec.normalized() -- the same operations on the same operands as the ones
// calc_rot_vec_c performs (every sequence is correctly rounded, so the bits are the same), which takes one
// sqrt / reciprocal / divide chain out of every latch: the Random heuristic's latch (59 of C2's 64 agents, every first
// contact of every rollout) is then one cross product.
template <int MATH = MATH_IEEE>
__device__ __forceinline__ V3 calc_rot_vec_pre(int type, V3 agent_pos, int n_obs, V3 own_pos, V3 closest_pos,
                                               V3 rand_vec, V3 to_obs, V3 goal_vec, double goal_dist, V3 goal_dir) {
  typedef Mth<MATH> M;
  if (type == T_GOAL || type == T_VEL) return mk(0.0, 0.0, 1.0);
  if (type == T_OBST) {
    if (n_obs < 2) return mk(0.0, 0.0, 1.0);
    V3 obstacle_vec = closest_pos - own_pos;
    V3 cur = to_obs * dot(obstacle_vec, to_obs) - obstacle_vec;
    return M::normalized(cross(cur, to_obs));
  }
  if (type == T_GOALOBST) {
    V3 obstacle_vec = closest_pos - own_pos;
    V3 obst_cur = to_obs * dot(obstacle_vec, to_obs) - obstacle_vec;
    V3 goal_cur = goal_vec - to_obs * dot(to_obs, goal_vec);
    V3 cur = M::normalized(goal_cur) + M::normalized(obst_cur);
    double s;
    V3 u;
    M::norm_unit(cur, s, u);
    cur = (s < 1e-10) ? mk(0.0, 0.0, 1.0) : u;
    return M::normalized(cross(cur, to_obs));
  }
  if (type == T_RANDOM) return cross(goal_dir, rand_vec);
  if (type == T_HAD) {
    V3 rob_obs = own_pos - agent_pos;
    V3 d = (agent_pos + goal_vec * M::div(dot(rob_obs, goal_vec), goal_dist * goal_dist)) - own_pos;
    V3 c = cross(d, goal_vec);
    return M::div3(c, M::norm(c));
  }
  return mk(0.0, 0.0, 0.0);
}

__device__ __forceinline__ V3 calc_rot_vec(int type, V3 agent_pos, V3 goal_pos, const ObsTab &T,
                                           int n_obs, int id, V3 own_pos, V3 rand_vec) {
  V3 closest_pos = own_pos;
  if ((type == T_OBST && n_obs >= 2) || type == T_GOALOBST) closest_pos = T.pos(closest_other(T, n_obs, id, own_pos));
  return calc_rot_vec_c(type, agent_pos, goal_pos, n_obs, own_pos, closest_pos, rand_vec);
}

// ---- circForce + attractorForceScaling over the group's lanes ------------
// act: this lane's agent takes a step AND its gate is open (uniform in the
// group). Must be called by all 64 lanes of the wave.
// rot_g / rand_g: this agent's rotation / random vectors in global memory,
// component-major [3][n_obs]. known_bits: bit t <-> obstacle t*LPA+sub.
template <int LPA, bool REAL>
__device__ __forceinline__ void circ_and_scale(bool act, int sub, int grp, int type, V3 p, V3 v,
                                               V3 goal, V3 g, const PopConst &C, double k_circ,
                                               const ObsTab &T, int n_obs, double *rot_g,
                                               const double *rand_g, unsigned long long &known_bits,
                                               double &min_obs, V3 &F, double &scale) {
  if (!__any(act)) return;
  const int M = n_obs - 1;
  const int ntiles = (M + LPA - 1) / LPA;
  V3 gn = normalized(g);
  double lane_min = min_obs;
  double best_d = C.shell;
  int best_i = 0x7fffffff;
  for (int t = 0; t < ntiles; t++) {
    int i = t * LPA + sub;
    bool valid = act && (i < M);
    int ii = valid ? i : 0;
    V3 op = T.pos(ii);
    V3 ov = T.vel(ii);
    double orad = T.r[ii];
    V3 ro = op - p;
    V3 rv = v - ov;
    double z = sqn(ro);
    double s = __builtin_sqrt(z);
    V3 ron = (z > 0.0) ? (ro / s) : ro;
    bool skip = (dot(ron, gn) < -0.01) && (dot(ro, rv) < -0.01);
    double d = s - (C.rad + orad);
    d = smax(d, 1e-5);
    // attractorForceScaling's sweep (:201-211) ignores the skip test
    if (valid && d < best_d) { best_d = d; best_i = i; }
    V3 c = mk(0.0, 0.0, 0.0);
    bool has_c = false;
    if (valid && !skip) {
      if (!REAL) {
        if (d < lane_min) lane_min = d;
      }
      if (d < C.shell) {
        V3 rot;
        if (!((known_bits >> t) & 1ull)) {
          V3 rnd = mk(0.0, 0.0, 0.0);
          if (type == T_RANDOM) rnd = mk(rand_g[i], rand_g[n_obs + i], rand_g[2 * n_obs + i]);
          rot = calc_rot_vec(type, p, goal, T, n_obs, i, op, rnd);
          rot_g[i] = rot.x; rot_g[n_obs + i] = rot.y; rot_g[2 * n_obs + i] = rot.z;
          known_bits |= (1ull << t);
        } else {
          rot = mk(rot_g[i], rot_g[n_obs + i], rot_g[2 * n_obs + i]);
        }
        double vn = norm(rv);
        if (vn != 0) {
          V3 nv = rv / vn;
          V3 cur = current_vector(type, rv, g, ron, rot);
          c = (k_circ / (d * d)) * cross(nv, cross(cur, nv));
          has_c = true;
        }
      }
    }
    ordered_accumulate<LPA>(F, c, has_c, grp);
  }
  if (!REAL) {
    lane_min = group_min<LPA>(lane_min);
    if (act) min_obs = lane_min;
  }
  group_argmin<LPA>(best_d, best_i);
  // attractorForceScaling, B/src/cf_agent.cpp:195-227 (only if |F| > 1e-5, :319)
  if (act && norm(F) > 1e-5) {
    if (best_i == 0x7fffffff) {
      scale = 1;
    } else if (dot(g, v) <= 0.0 && norm(v) < C.vel_max - 0.1 * C.vel_max && norm(g) > 0.15) {
      scale = 0.0;
    } else {
      double w1 = 1 - portable_exp(-__builtin_sqrt(best_d) / C.shell);
      V3 ro = T.pos(best_i) - p;
      double w2 = 1 - (dot(g, ro) / (norm(g) * norm(ro)));
      w2 = w2 * w2;
      scale = w1 * w2;
    }
  }
}

// repelForce (:159-181) + attractorForce (:183-193) + updatePositionAndVelocity
// (:253-268): O(1) per agent, evaluated by every lane of the group.
__device__ __forceinline__ void finish_step(V3 p, V3 &v, V3 goal_vec, V3 &F, double scale,
                                            const PopConst &C, double k_attr, double k_repel,
                                            double k_damp, double dt, V3 sent_pos, double sent_rad,
                                            V3 &new_pos) {
  {
    V3 ro = sent_pos - p;
    V3 dist_vec = -ro;
    double d = norm(dist_vec) - (C.rad + sent_rad);
    d = smax(d, 1e-5);
    V3 repel = mk(0.0, 0.0, 0.0);
    if (d < C.shell) {
      V3 otr = normalized(p - sent_pos);
      double t = 1.0 / d - 1.0 / C.shell;
      double dd = d * d;
      repel = ((k_repel * otr) * t) / dd;
    }
    V3 total = mk(0.0, 0.0, 0.0) + repel;
    F = F + total;
  }
  if (k_attr != 0.0) {
    V3 vel_des = (k_attr / k_damp) * goal_vec;
    double scale_lim = smin(1.0, C.vel_max / norm(vel_des));
    vel_des = vel_des * scale_lim;
    F = F + (scale * k_damp) * (vel_des - v);
  }
  // force_ / mass_; x / 1.0 == x exactly for every x, so the reference's
  // default mass (1.0) needs no division
  V3 acc = F;
  if (C.mass != 1.0) acc = F / C.mass;
  double an = norm(acc);
  if (an > 13.0) acc = acc * (13.0 / an);
  V3 half = ((0.5 * acc) * dt) * dt;
  new_pos = (p + half) + (v * dt);
  V3 nv = v + acc * dt;
  double vn = norm(nv);
  if (vn > C.vel_max) nv = nv * (C.vel_max / vn);
  v = nv;
}

// workspace-box penalty of one path point, B/src/cf_manager.cpp:302-324;
// adds up to three terms to cost in x,y,z order.
__device__ __forceinline__ void ws_cost_add(double &cost, V3 q, const double *ws, double k_ws) {
  // one (rarely taken) branch for the common in-box case
  const bool out = (q.x > ws[0]) | (q.x < ws[1]) | (q.y > ws[2]) | (q.y < ws[3]) | (q.z > ws[4]) | (q.z < ws[5]);
  if (out) {
    double t;
    if (q.x > ws[0]) { t = fabs(q.x - ws[0]) * k_ws; cost += t * t; }
    else if (q.x < ws[1]) { t = fabs(q.x - ws[1]) * k_ws; cost += t * t; }
    if (q.y > ws[2]) { t = fabs(q.y - ws[2]) * k_ws; cost += t * t; }
    else if (q.y < ws[3]) { t = fabs(q.y - ws[3]) * k_ws; cost += t * t; }
    if (q.z > ws[4]) { t = fabs(q.z - ws[4]) * k_ws; cost += t * t; }
    else if (q.z < ws[5]) { t = fabs(q.z - ws[5]) * k_ws; cost += t * t; }
  }
}

}  // namespace pmaf

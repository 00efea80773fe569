// flopcount.cpp -- TEST INFRASTRUCTURE ONLY (not product code).
//
// The CPU restatement (pmaf_oracle.c) compiled a second time, as C++, with `double` replaced by an instrumented
// scalar type that counts every floating-point operation it executes: + - * / sqrt exp and comparisons count 1
// each (SURVEY.md 8d's convention; negation, fabs and copies count 0). bench.py uses it -- as a measuring device
// only -- to report the exact FP64 operation count per agent-step of its workload next to the throughput
// ("flops_per_agent_step_measured"), and the fraction of agent-steps with at least one in-shell obstacle.
// System headers are included BEFORE the macro below so that only the oracle's own code sees the counting type.
#include <math.h>

#include <cmath>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static long long g_flops = 0;

struct cdouble {
  double v;
  cdouble() = default;
  cdouble(double x) : v(x) {}
  explicit operator int() const { return (int)v; }
  explicit operator double() const { return v; }
  cdouble operator-() const { return cdouble(-v); }
  cdouble &operator+=(cdouble o) { g_flops++; v += o.v; return *this; }
};
#define PMAF_BIN(op)                                                                       \
  static inline cdouble operator op(cdouble a, cdouble b) { g_flops++; return cdouble(a.v op b.v); } \
  static inline cdouble operator op(cdouble a, double b) { g_flops++; return cdouble(a.v op b); }    \
  static inline cdouble operator op(double a, cdouble b) { g_flops++; return cdouble(a op b.v); }
PMAF_BIN(+) PMAF_BIN(-) PMAF_BIN(*) PMAF_BIN(/)
#undef PMAF_BIN
#define PMAF_CMP(op)                                                                  \
  static inline bool operator op(cdouble a, cdouble b) { g_flops++; return a.v op b.v; } \
  static inline bool operator op(cdouble a, double b) { g_flops++; return a.v op b; }    \
  static inline bool operator op(double a, cdouble b) { g_flops++; return a op b.v; }
PMAF_CMP(<) PMAF_CMP(>) PMAF_CMP(<=) PMAF_CMP(>=) PMAF_CMP(==) PMAF_CMP(!=)
#undef PMAF_CMP
static inline cdouble sqrt(cdouble a) { g_flops++; return cdouble(sqrt(a.v)); }
static inline cdouble exp(cdouble a) { g_flops++; return cdouble(exp(a.v)); }
static inline cdouble fabs(cdouble a) { return cdouble(fabs(a.v)); }
static inline bool isnan(cdouble a) { return std::isnan(a.v); }
// pmaf_portable_exp's operations (exp mode 1; the count uses mode 0, where exp() is ONE operation): they only have to compile
static inline cdouble fma(cdouble a, cdouble b, cdouble c) { g_flops += 2; return cdouble(fma(a.v, b.v, c.v)); }
static inline cdouble rint(cdouble a) { g_flops++; return cdouble(rint(a.v)); }
static inline cdouble ldexp(cdouble a, int k) { g_flops++; return cdouble(ldexp(a.v, k)); }
static inline cdouble fmin(cdouble a, cdouble b) { g_flops++; return cdouble(fmin(a.v, b.v)); }
static inline cdouble fmax(cdouble a, cdouble b) { g_flops++; return cdouble(fmax(a.v, b.v)); }

#define PMAF_FLOPCOUNT 1
#define double cdouble
#include "pmaf_oracle.c"
#undef double

extern "C" {
long long orc_flops(void) { return g_flops; }
void orc_flops_reset(void) { g_flops = 0; g_steps_total = 0; g_steps_in_shell = 0; }
long long orc_steps_total(void) { return g_steps_total; }
long long orc_steps_in_shell(void) { return g_steps_in_shell; }
}

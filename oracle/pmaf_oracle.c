/*
 * pmaf_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * Scalar CPU restatement of the reference planner tick. PARITY UNPINNED by
 * the reference's own tests (it has none); see pmaf_oracle.h and DESIGN.md.
 * B/ = /root/reference/src/bimanual_planning_ros/.
 *
 * Floating-point evaluation order mirrors what the reference's Eigen
 * expressions evaluate to on x86-64:
 *   dot/squaredNorm of a 3-vector: (a0*b0 + a1*b1) + a2*b2  (Eigen 3.3
 *     linear-vectorised redux with a 2-wide packet + scalar tail; define
 *     PMAF_DOT_RIGHT_ASSOC for a0*b0 + (a1*b1 + a2*b2), the non-vectorised
 *     unrolled redux)
 *   norm = sqrt(squaredNorm); normalized(): divide each component by
 *     sqrt(z) if z > 0 else unchanged; cross = (a1b2-a2b1, a2b0-a0b2,
 *     a0b1-a1b0); pow(x,2) = x*x; std::max(a,b) = (a<b)?b:a;
 *     std::min(a,b) = (b<a)?b:a.
 * Build with -ffp-contract=off (no FMA fusion).
 */
#include "pmaf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vscale(double s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline v3 vmuls(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
#ifdef PMAF_QUOTIENT_BY_RECIPROCAL
/* sensitivity study only (tests/test_oracle_sensitivity.py): vector / scalar as multiplication by the reciprocal,
 * what an Eigen older than the reference's would evaluate */
static inline v3 vdivs(v3 a, double s) { double r = 1.0 / s; return V(a.x * r, a.y * r, a.z * r); }
#else
static inline v3 vdivs(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
#endif
static inline double vdot(v3 a, v3 b) {
#ifdef PMAF_DOT_RIGHT_ASSOC
  return a.x * b.x + (a.y * b.y + a.z * b.z);
#else
  return (a.x * b.x + a.y * b.y) + a.z * b.z;
#endif
}
/* which association this build uses: 0 = (a0 b0 + a1 b1) + a2 b2, 1 = a0 b0 + (a1 b1 + a2 b2) -- the values of
 * include/pmaf.h's PMAF_EVAL_ORDER_DOT_LEFT / _RIGHT; the parity suite refuses to compare a library with an oracle of
 * the other order (tests/conftest.py) */
int orc_eval_order(void) {
#ifdef PMAF_DOT_RIGHT_ASSOC
  return 1;
#else
  return 0;
#endif
}
static inline double vsqn(v3 a) { return vdot(a, a); }
static inline double vnorm(v3 a) { return sqrt(vsqn(a)); }
static inline v3 vnormalized(v3 a) {
  double z = vsqn(a);
  if (z > 0.0) return vdivs(a, sqrt(z));
  return a;
}
static inline v3 vcross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/*
 * exp(): the reference calls std::exp (B/src/cf_agent.cpp:220), i.e. the
 * platform libm, whose last bit is not the same in every libm. Mode 0 (default)
 * keeps the host's libm exp: the reference-faithful restatement. Mode 1 uses
 * pmaf_portable_exp below, a restatement of glibc >= 2.28's exp in correctly
 * rounded IEEE operations only (fma, *, +, integer bit operations, a table of
 * 2^(k/128)), so it gives the same bits on every IEEE platform; the HIP kernels
 * evaluate the same function (csrc/pmaf_device.hpp), which makes
 * kernel-vs-oracle comparisons bit-exact in mode 1 -- and on a host whose libm
 * IS that algorithm (any x86-64 glibc >= 2.28 on a CPU with FMA) mode 1 and
 * mode 0 are the same function.
 */
static int g_exp_mode = 0;
void orc_set_exp_mode(int mode) { g_exp_mode = mode; }
int orc_get_exp_mode(void) { return g_exp_mode; }

#ifdef PMAF_FLOPCOUNT
/* the operation-counting build (flopcount.cpp) counts in exp mode 0, where exp() is ONE operation */
double pmaf_portable_exp(double x) { return exp(x); }
#else
#include "pmaf_exp_table.h"
/* The exp() the reference calls (std::exp, B/src/cf_agent.cpp:220 = glibc's libm), RESTATED so that host and gfx950
 * evaluate the same operations: glibc >= 2.28's algorithm (sysdeps/ieee754/dbl-64/e_exp.c; the variant compiled with FMA
 * contraction, which the ifunc selects on every x86-64 CPU with FMA) -- N = 128 table of 2^(k/N), degree-5 polynomial,
 * operation order as in tools/gen_exp_table.py's header. Every operation is a correctly rounded IEEE one (fma() is
 * exact-then-rounded with or without hardware FMA), so the bits equal csrc/pmaf_device.hpp:portable_exp on gfx950 AND the
 * host libm's exp() wherever that libm is this algorithm (tests/test_oracle_properties.py checks it on the machine it runs
 * on: 2e8 arguments without a mismatch on the build image, glibc 2.35). Error 0.511 ulp.
 * Range: x is clamped to >= -500 -- below -512 glibc leaves this path for its subnormal-safe one (an unfused last step),
 * and the only caller forms 1 - exp(x), which is 1.0 for every exp(x) < 2^-54 (x < -37.4): the clamp changes no result
 * of the planner and keeps ONE straight-line sequence on the device. x >= 512: glibc's specialcase() for large positive
 * arguments, restated too (the exponent of the scale would overflow); x above 709.78 -> +inf like exp(). */
double pmaf_portable_exp(double x) {
  double kd, r, tail, scale;
  uint64_t ki, sbits, u;
  const double *K = (const double *)(const void *)PMAF_EXP_DATA;   /* the eight constants' bit patterns as doubles */
  if (x != x) return x;                                    /* NaN */
  if (x > 0x1.62e42fefa39efp+9) return INFINITY;           /* overflow threshold of exp() */
  const double xs = fmax(x, -500.0);
  kd = fma(xs, K[0], K[1]);
  memcpy(&ki, &kd, 8);
  kd = kd - K[1];
  r = fma(kd, K[2], xs);
  r = fma(kd, K[3], r);
  u = PMAF_EXP_DATA[8 + 2 * (ki & (PMAF_EXP_N - 1))];
  memcpy(&tail, &u, 8);
  sbits = PMAF_EXP_DATA[9 + 2 * (ki & (PMAF_EXP_N - 1))] + (ki << 45);
  {
    const double a = fma(K[5], r, K[4]);                   /* C2 + r C3 */
    const double tr = r + tail;
    const double r2 = r * r;
    const double b = fma(r, K[7], K[6]);                   /* C4 + r C5 */
    const double c = fma(a, r2, tr);
    const double r4 = r2 * r2;
    const double tmp = fma(r4, b, c);
    if (xs >= 512.0) {
      /* glibc's specialcase(), k > 0 branch: up here 2^(k/N)'s exponent field may overflow (k / N reaches 1024 just
       * below the threshold: `sbits` would be +inf's pattern and fma(inf, tmp < 0, inf) a NaN -- ADVICE r5), so the scale
       * is formed 2^-1009 lower and the result multiplied back up; above the threshold the product overflows to +inf by
       * itself. The planner never gets here (its arguments are <= 0). */
      sbits -= (uint64_t)1009 << 52;
      memcpy(&scale, &sbits, 8);
      return 0x1p1009 * fma(scale, tmp, scale);
    }
    memcpy(&scale, &sbits, 8);
    return fma(scale, tmp, scale);
  }
}
#endif
static inline double orc_exp(double x) { return g_exp_mode ? pmaf_portable_exp(x) : exp(x); }
/* array forms for the tests that compare the restatement with the host libm (and the kernels with both) on millions of
 * arguments: which = 1 -> pmaf_portable_exp, 0 -> the host libm's exp() */
void pmaf_exp_array(int which, const double *x, double *y, long n) {
  for (long i = 0; i < n; i++) y[i] = which ? pmaf_portable_exp(x[i]) : exp(x[i]);
}

static inline double dmax(double a, double b) { return (a < b) ? b : a; }
static inline double dmin(double a, double b) { return (b < a) ? b : a; }

/* Obstacle value type, B/include/bimanual_planning_ros/obstacle.h:16-40 */
typedef struct { v3 pos, vel; double rad; } obs_t;

/* CfAgent state, B/include/bimanual_planning_ros/cf_agent.h:36-56 */
typedef struct {
  int id;              /* 1-based */
  int type;
  v3 *pos;             /* path, capacity cap */
  int n_pos;
  int cap;
  v3 vel, init_pos, g_pos, force;
  double shell, mass, rad, vel_max, min_obs_dist, approach;
  obs_t *obstacles;    /* private copy, n_obs */
  unsigned char *known;
  v3 *rot;             /* field_rotation_vecs_ */
  const v3 *rand_vecs; /* RandomCfAgent::random_vecs_ (n_obs) or NULL */
  int reached_goal;
} agent_t;

struct orc_planner {
  int n_agents, n_obs, cap;
  double dt;
  v3 goal, init_pos;
  double approach;
  double *k_attr, *k_circ, *k_repel, *k_damp;
  agent_t *agents;
  agent_t real;        /* RealCfAgent; path grows without bound */
  int real_cap;
  int has_best;        /* best_agent_ (copy), cf_manager.h:20 */
  int best_id, best_type;
  v3 *best_rand;
  v3 *rand_all;        /* [n_agents][n_obs] */
  double *costs;
  int64_t agent_steps;
};

#ifdef PMAF_FLOPCOUNT
/* flopcount.cpp: agent-steps of the rollouts, and those with at least one obstacle inside the detection shell */
static long long g_steps_total = 0, g_steps_in_shell = 0;
static int g_step_in_shell = 0;
#endif

static v3 latest(const agent_t *a) { return a->pos[a->n_pos - 1]; }
/* CfAgent::getDistFromGoal, cf_agent.h:116-118 */
static double dist_goal(const agent_t *a) { return vnorm(vsub(a->g_pos, latest(a))); }

static void push_pos(agent_t *a, v3 p) {
  if (a->n_pos == a->cap) {           /* only the real agent can grow */
    a->cap = a->cap * 2 + 16;
    a->pos = (v3 *)realloc(a->pos, sizeof(v3) * (size_t)a->cap);
  }
  a->pos[a->n_pos++] = p;
}

/* ------------------------------------------------------------------ */
/* currentVector per heuristic, B/src/cf_agent.cpp:389-406, 414-426,
 * 463-475, 520-537, 545-557, 585-597 */
static v3 current_vector(int type, v3 agent_pos, v3 agent_vel, v3 goal_pos,
                         const obs_t *obstacles, int id, const v3 *rot) {
  switch (type) {
    case ORC_GOAL_HEURISTIC: {           /* :389-406 */
      v3 goal_vec = vsub(goal_pos, agent_pos);
      v3 to_obs = vnormalized(vsub(obstacles[id].pos, agent_pos));
      v3 cur = vsub(goal_vec, vmuls(to_obs, vdot(to_obs, goal_vec)));
      if (vnorm(cur) < 1e-10) cur = V(0.0, 0.0, 1.0);
      return vnormalized(cur);
    }
    case ORC_VEL_HEURISTIC: {            /* :520-537 */
      v3 nvel = vnormalized(agent_vel);
      v3 to_obs = vnormalized(vsub(obstacles[id].pos, agent_pos));
      v3 cur = vsub(nvel, vmuls(to_obs, vdot(nvel, to_obs)));
      if (vnorm(cur) < 1e-10) cur = V(0.0, 0.0, 1.0);
      return vnormalized(cur);
    }
    case ORC_OBSTACLE_HEURISTIC:         /* :414-426 */
    case ORC_GOAL_OBSTACLE_HEURISTIC:    /* :463-475 */
    case ORC_RANDOM_AGENT:               /* :545-557 */
    case ORC_HAD_HEURISTIC: {            /* :585-597 */
      v3 to_obs = vnormalized(vsub(obstacles[id].pos, agent_pos));
      return vnormalized(vcross(to_obs, rot[id]));
    }
    default:
      /* base-class body is empty (cf_agent.h:155-159): undefined in the
       * reference; the restatement returns zero. */
      return V(0.0, 0.0, 0.0);
  }
}

/* nearest *other* field obstacle by centre distance, B/src/cf_agent.cpp:434-446
 * and :480-492 */
static int closest_other(const obs_t *obstacles, int n_obs, int id) {
  double min_dist = 100.0;
  int closest = 0;
  for (int i = 0; i < n_obs - 1; i++) {
    if (i != id) {
      double d = vnorm(vsub(obstacles[id].pos, obstacles[i].pos));
      if (min_dist > d) { min_dist = d; closest = i; }
    }
  }
  return closest;
}

/* calculateRotationVector per heuristic, B/src/cf_agent.cpp:408-412, 428-461,
 * 477-518, 539-543, 559-566, 599-611 */
static v3 calc_rot_vec(int type, v3 agent_pos, v3 goal_pos,
                       const obs_t *obstacles, int n_obs, int id,
                       const v3 *rand_vecs) {
  switch (type) {
    case ORC_GOAL_HEURISTIC:             /* :408-412 */
    case ORC_VEL_HEURISTIC:              /* :539-543 */
      return V(0.0, 0.0, 1.0);
    case ORC_OBSTACLE_HEURISTIC: {       /* :428-461 */
      if (n_obs < 2) return V(0.0, 0.0, 1.0);
      int c = closest_other(obstacles, n_obs, id);
      v3 obstacle_vec = vsub(obstacles[c].pos, obstacles[id].pos);
      v3 to_obs = vnormalized(vsub(obstacles[id].pos, agent_pos));
      v3 cur = vsub(vmuls(to_obs, vdot(obstacle_vec, to_obs)), obstacle_vec);
      return vnormalized(vcross(cur, to_obs));
    }
    case ORC_GOAL_OBSTACLE_HEURISTIC: {  /* :477-518 */
      int c = closest_other(obstacles, n_obs, id);
      v3 obstacle_vec = vsub(obstacles[c].pos, obstacles[id].pos);
      v3 to_obs = vnormalized(vsub(obstacles[id].pos, agent_pos));
      v3 obst_cur = vsub(vmuls(to_obs, vdot(obstacle_vec, to_obs)), obstacle_vec);
      v3 goal_vec = vsub(goal_pos, agent_pos);
      v3 goal_cur = vsub(goal_vec, vmuls(to_obs, vdot(to_obs, goal_vec)));
      v3 cur = vadd(vnormalized(goal_cur), vnormalized(obst_cur));
      if (vnorm(cur) < 1e-10) cur = V(0.0, 0.0, 1.0);
      cur = vnormalized(cur);
      return vnormalized(vcross(cur, to_obs));
    }
    case ORC_RANDOM_AGENT: {             /* :559-566, not normalised */
      v3 goal_vec = vnormalized(vsub(goal_pos, agent_pos));
      return vcross(goal_vec, rand_vecs[id]);
    }
    case ORC_HAD_HEURISTIC: {            /* :599-611, unguarded division */
      v3 obs_pos = obstacles[id].pos;
      v3 goal_vec = vsub(goal_pos, agent_pos);
      v3 rob_obs = vsub(obs_pos, agent_pos);
      double gn = vnorm(goal_vec);
      v3 d = vsub(vadd(agent_pos, vmuls(goal_vec, vdot(rob_obs, goal_vec) / (gn * gn))), obs_pos);
      v3 c = vcross(d, goal_vec);
      return vdivs(c, vnorm(c));
    }
    default:
      return V(0.0, 0.0, 0.0);
  }
}

/* ------------------------------------------------------------------ */
/* CfAgent::circForce, B/src/cf_agent.cpp:72-108 (track_min = 1), and
 * RealCfAgent::circForce, :110-144 (track_min = 0, heuristics dispatched to
 * the best agent: htype / hrand). */
static void circ_force(agent_t *a, const obs_t *obstacles, int n_obs,
                       double k_circ, int track_min, int htype,
                       const v3 *hrand) {
  v3 p = latest(a);
  v3 goal_vec = vsub(a->g_pos, p);
  for (int i = 0; i < n_obs - 1; i++) {
    v3 ro = vsub(obstacles[i].pos, p);
    v3 rel_vel = vsub(a->vel, obstacles[i].vel);
    if (vdot(vnormalized(ro), vnormalized(goal_vec)) < -0.01 &&
        vdot(ro, rel_vel) < -0.01) {
      continue;
    }
    /* :83-84 uses robot_obstacle_vec.norm(); :121-123 (p - o).norm():
     * identical bits (squares of negated components). */
    double dist_obs = vnorm(ro) - (a->rad + obstacles[i].rad);
    dist_obs = dmax(dist_obs, 1e-5);
    if (track_min && dist_obs < a->min_obs_dist) a->min_obs_dist = dist_obs;
    v3 curr_force = V(0.0, 0.0, 0.0);
    if (dist_obs < a->shell) {
#ifdef PMAF_FLOPCOUNT
      g_step_in_shell = 1;
#endif
      if (!a->known[i]) {
        a->rot[i] = calc_rot_vec(htype, p, a->g_pos, obstacles, n_obs, i, hrand);
        a->known[i] = 1;
      }
      double vel_norm = vnorm(rel_vel);
      if (vel_norm != 0) {
        v3 nvel = vdivs(rel_vel, vel_norm);
        v3 cur = current_vector(htype, p, rel_vel, a->g_pos, obstacles, i, a->rot);
        curr_force = vscale(k_circ / (dist_obs * dist_obs), vcross(nvel, vcross(cur, nvel)));
      }
    }
    a->force = vadd(a->force, curr_force);
  }
}

/* CfAgent::repelForce, B/src/cf_agent.cpp:159-181 */
static void repel_force(agent_t *a, const obs_t *obstacles, int n_obs, double k_repel) {
  const obs_t *o = &obstacles[n_obs - 1];
  v3 p = latest(a);
  v3 ro = vsub(o->pos, p);
  v3 dist_vec = vneg(ro);
  double dist_obs = vnorm(dist_vec) - (a->rad + o->rad);
  dist_obs = dmax(dist_obs, 1e-5);
  v3 repel = V(0.0, 0.0, 0.0);
  if (dist_obs < a->shell) {
    v3 obs_to_robot = vnormalized(vsub(p, o->pos));
    /* k_repel * vec * (1/d - 1/shell) / (d*d), evaluated left to right */
    double t = 1.0 / dist_obs - 1.0 / a->shell;
    double dd = dist_obs * dist_obs;
    v3 kv = vscale(k_repel, obs_to_robot);
    repel = vdivs(vmuls(kv, t), dd);
  }
  v3 total = vadd(V(0.0, 0.0, 0.0), repel);
  a->force = vadd(a->force, total);
}

/* CfAgent::attractorForce, B/src/cf_agent.cpp:183-193 */
static void attractor_force(agent_t *a, double k_attr, double k_damp, double k_goal_scale) {
  if (k_attr == 0.0) return;
  v3 goal_vec = vsub(a->g_pos, latest(a));
  v3 vel_des = vscale(k_attr / k_damp, goal_vec);
  double scale_lim = dmin(1.0, a->vel_max / vnorm(vel_des));
  vel_des = vmuls(vel_des, scale_lim);
  a->force = vadd(a->force, vscale(k_goal_scale * k_damp, vsub(vel_des, a->vel)));
}

/* CfAgent::attractorForceScaling, B/src/cf_agent.cpp:195-227 */
static double attractor_force_scaling(const agent_t *a, const obs_t *obstacles, int n_obs) {
  int id_closest = 0;
  int no_close = 1;
  double closest = a->shell;
  v3 p = latest(a);
  for (int i = 0; i < n_obs - 1; i++) {
    double d = vnorm(vsub(p, obstacles[i].pos)) - (a->rad + obstacles[i].rad);
    d = dmax(d, 1e-5);
    if (d < closest) { no_close = 0; closest = d; id_closest = i; }
  }
  if (no_close) return 1;
  v3 goal_vec = vsub(a->g_pos, p);
  if (vdot(goal_vec, a->vel) <= 0.0 && vnorm(a->vel) < a->vel_max - 0.1 * a->vel_max &&
      vnorm(goal_vec) > 0.15) {
    return 0.0;
  }
  double w1 = 1 - orc_exp(-sqrt(closest) / a->shell);
  v3 ro = vsub(obstacles[id_closest].pos, p);
  double w2 = 1 - (vdot(goal_vec, ro) / (vnorm(goal_vec) * vnorm(ro)));
  w2 = w2 * w2;
  return w1 * w2;
}

/* CfAgent::updatePositionAndVelocity, B/src/cf_agent.cpp:253-268 */
static void update_pos_vel(agent_t *a, double dt) {
  v3 acc = vdivs(a->force, a->mass);
  double acc_norm = vnorm(acc);
  if (acc_norm > 13.0) acc = vmuls(acc, 13.0 / acc_norm);
  /* p + 0.5*a*dt*dt + v*dt  ==  (p + ((0.5*a)*dt)*dt) + v*dt */
  v3 half = vmuls(vmuls(vscale(0.5, acc), dt), dt);
  v3 new_pos = vadd(vadd(latest(a), half), vmuls(a->vel, dt));
  a->vel = vadd(a->vel, vmuls(acc, dt));
  double vel_norm = vnorm(a->vel);
  if (vel_norm > a->vel_max) a->vel = vmuls(a->vel, a->vel_max / vel_norm);
  push_pos(a, new_pos);
}

/* CfAgent::predictObstacles, B/src/cf_agent.cpp:270-276 */
static void predict_obstacles(agent_t *a, int n_obs, double dt) {
  for (int i = 0; i < n_obs; i++) {
    a->obstacles[i].pos = vadd(a->obstacles[i].pos, vmuls(a->obstacles[i].vel, dt));
  }
}

/* gate, B/src/cf_agent.cpp:315-317 (and :287-289, :352-354) */
static int gate_open(const agent_t *a) {
  return !(dist_goal(a) < a->approach ||
           (vnorm(a->vel) < 0.5 * a->vel_max &&
            vnorm(vsub(latest(a), a->init_pos)) < 0.2));
}

/* one body of the cfPrediction loop, B/src/cf_agent.cpp:312-327 */
static void prediction_step(agent_t *a, int n_obs, double k_attr, double k_circ,
                            double k_repel, double k_damp, double dt) {
  a->force = V(0.0, 0.0, 0.0);
  double k_goal_scale = 1.0;
#ifdef PMAF_FLOPCOUNT
  g_step_in_shell = 0;
#endif
  if (gate_open(a)) {
    circ_force(a, a->obstacles, n_obs, k_circ, 1, a->type, a->rand_vecs);
    if (vnorm(a->force) > 1e-5) {
      k_goal_scale = attractor_force_scaling(a, a->obstacles, n_obs);
    }
  }
  repel_force(a, a->obstacles, n_obs, k_repel);
  attractor_force(a, k_attr, k_damp, k_goal_scale);
  update_pos_vel(a, dt);
  predict_obstacles(a, n_obs, dt);
#ifdef PMAF_FLOPCOUNT
  g_steps_total++;
  g_steps_in_shell += g_step_in_shell;
#endif
}

/* CfAgent::cfPrediction inner loop run until its guard fails,
 * B/src/cf_agent.cpp:310-337 */
static int64_t run_prediction(orc_planner *p, int i) {
  agent_t *a = &p->agents[i];
  int ran = 0;
  int64_t steps = 0;
  while (dist_goal(a) > 0.1 && a->n_pos < p->cap) {
    ran = 1;
    prediction_step(a, p->n_obs, p->k_attr[i], p->k_circ[i], p->k_repel[i],
                    p->k_damp[i], p->dt);
    steps++;
  }
  if (ran) a->reached_goal = dist_goal(a) < 0.100001;
  return steps;
}

/* ------------------------------------------------------------------ */
static void agent_init(agent_t *a, int id, int type, v3 pos, v3 goal,
                       double shell, double mass, double rad, double vel_max,
                       double approach, int n_obs, const obs_t *obstacles,
                       int cap, const v3 *rand_vecs) {
  /* CfAgent ctor, cf_agent.h:69-97 */
  memset(a, 0, sizeof(*a));
  a->id = id;
  a->type = type;
  a->cap = cap;
  a->pos = (v3 *)malloc(sizeof(v3) * (size_t)cap);
  a->pos[0] = pos;
  a->n_pos = 1;
  a->vel = V(0.01, 0.0, 0.0);
  a->init_pos = V(0.0, 0.0, 0.0);
  a->g_pos = goal;
  a->force = V(0.0, 0.0, 0.0);
  a->shell = shell;
  a->min_obs_dist = shell;
  a->mass = mass;
  a->rad = rad;
  a->vel_max = vel_max;
  a->approach = approach;
  a->reached_goal = 0;
  a->known = (unsigned char *)calloc((size_t)n_obs, 1);
  a->rot = (v3 *)malloc(sizeof(v3) * (size_t)n_obs);
  for (int i = 0; i < n_obs; i++) a->rot[i] = V(0.0, 0.0, 1.0);
  if (obstacles) {
    a->obstacles = (obs_t *)malloc(sizeof(obs_t) * (size_t)n_obs);
    memcpy(a->obstacles, obstacles, sizeof(obs_t) * (size_t)n_obs);
  }
  a->rand_vecs = rand_vecs;
}

static void agent_free(agent_t *a) {
  free(a->pos); free(a->known); free(a->rot); free(a->obstacles);
}

static void unpack_obstacles(const double *flat, int n_obs, obs_t *out) {
  for (int i = 0; i < n_obs; i++) {
    out[i].pos = V(flat[7 * i + 0], flat[7 * i + 1], flat[7 * i + 2]);
    out[i].vel = V(flat[7 * i + 3], flat[7 * i + 4], flat[7 * i + 5]);
    out[i].rad = flat[7 * i + 6];
  }
}

orc_planner *orc_create(int n_agents, int n_obs, int max_prediction_steps,
                        const double *scal, const double *goal,
                        const double *mgr_init_pos, const double *obstacles,
                        const double *gains, const int32_t *types,
                        const double *random_vecs) {
  if (n_agents < 1 || n_obs < 1 || max_prediction_steps < 1) return NULL;
  orc_planner *p = (orc_planner *)calloc(1, sizeof(*p));
  p->n_agents = n_agents;
  p->n_obs = n_obs;
  p->cap = max_prediction_steps;
  p->dt = scal[0];
  double vel_max = scal[1], approach = scal[2], shell = scal[3], mass = scal[4], rad = scal[5];
  p->approach = approach;
  p->goal = V(goal[0], goal[1], goal[2]);
  p->init_pos = V(mgr_init_pos[0], mgr_init_pos[1], mgr_init_pos[2]);
  size_t nb = sizeof(double) * (size_t)n_agents;
  p->k_attr = (double *)malloc(nb); memcpy(p->k_attr, gains + 0 * n_agents, nb);
  p->k_circ = (double *)malloc(nb); memcpy(p->k_circ, gains + 1 * n_agents, nb);
  p->k_repel = (double *)malloc(nb); memcpy(p->k_repel, gains + 2 * n_agents, nb);
  p->k_damp = (double *)malloc(nb); memcpy(p->k_damp, gains + 3 * n_agents, nb);
  p->costs = (double *)calloc((size_t)n_agents, sizeof(double));
  obs_t *obs = (obs_t *)malloc(sizeof(obs_t) * (size_t)n_obs);
  unpack_obstacles(obstacles, n_obs, obs);
  p->rand_all = (v3 *)calloc((size_t)n_agents * (size_t)n_obs, sizeof(v3));
  if (random_vecs) memcpy(p->rand_all, random_vecs, sizeof(v3) * (size_t)n_agents * (size_t)n_obs);
  p->best_rand = (v3 *)calloc((size_t)n_obs, sizeof(v3));
  /* real agent, cf_manager.cpp:66-68; it owns no obstacle copy */
  agent_init(&p->real, 0, ORC_REAL_AGENT, p->init_pos, p->goal, shell, mass, rad,
             vel_max, approach, n_obs, NULL, 64, NULL);
  /* population layout, cf_manager.cpp:70-104 */
  static const int layout[5] = {ORC_HAD_HEURISTIC, ORC_GOAL_HEURISTIC,
                                ORC_OBSTACLE_HEURISTIC,
                                ORC_GOAL_OBSTACLE_HEURISTIC, ORC_VEL_HEURISTIC};
  p->agents = (agent_t *)calloc((size_t)n_agents, sizeof(agent_t));
  for (int i = 0; i < n_agents; i++) {
    int type = types ? types[i] : (i < 5 ? layout[i] : ORC_RANDOM_AGENT);
    agent_init(&p->agents[i], i + 1, type, p->init_pos, p->goal, shell, mass, rad,
               vel_max, approach, n_obs, obs, p->cap,
               p->rand_all + (size_t)i * (size_t)n_obs);
  }
  free(obs);
  return p;
}

void orc_destroy(orc_planner *p) {
  if (!p) return;
  for (int i = 0; i < p->n_agents; i++) agent_free(&p->agents[i]);
  agent_free(&p->real);
  free(p->agents); free(p->k_attr); free(p->k_circ); free(p->k_repel);
  free(p->k_damp); free(p->costs); free(p->rand_all); free(p->best_rand);
  free(p);
}

/* CfManager::setInitialPosition -> setInitialEEPositions -> setInitalPosition,
 * B/src/cf_manager.cpp:226-236, B/src/cf_agent.cpp:34-46 */
void orc_set_initial_position(orc_planner *p, const double *pos) {
  v3 q = V(pos[0], pos[1], pos[2]);
  p->init_pos = q;
  p->real.init_pos = q;
  push_pos(&p->real, q);            /* RealCfAgent::setPosition = push_back */
  for (int i = 0; i < p->n_agents; i++) {
    p->agents[i].init_pos = q;
    p->agents[i].n_pos = 0;         /* CfAgent::setPosition = clear + push_back */
    push_pos(&p->agents[i], q);
  }
}

void orc_set_real_position(orc_planner *p, const double *pos) {
  push_pos(&p->real, V(pos[0], pos[1], pos[2]));
}

void orc_rollout_range(orc_planner *p, int a0, int a1) {
  int64_t steps = 0;
  for (int i = a0; i < a1; i++) steps += run_prediction(p, i);
  __atomic_fetch_add(&p->agent_steps, steps, __ATOMIC_RELAXED);
}

void orc_rollout(orc_planner *p) { orc_rollout_range(p, 0, p->n_agents); }

/* all agents' rollouts on n_threads OpenMP threads (agents are independent,
 * like the reference's thread per agent, B/src/cf_manager.cpp:118-123); used
 * by the multi-core CPU baseline of bench.py */
void orc_rollout_omp(orc_planner *p, int n_threads) {
  int64_t steps = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : steps)
  for (int i = 0; i < p->n_agents; i++) steps += run_prediction(p, i);
  p->agent_steps += steps;
}

/* planCallback sequence with the multi-threaded rollout */
int orc_tick_omp(orc_planner *p, const double *obstacles, double dt, const double *cost_gains, const double *ws,
                 int n_threads) {
  int best = orc_evaluate(p, cost_gains[0], cost_gains[1], cost_gains[2], cost_gains[3], ws);
  orc_move_real(p, obstacles, dt, 1, best);
  v3 np = latest(&p->real);
  double pos[3] = {np.x, np.y, np.z};
  double vel[3] = {p->real.vel.x, p->real.vel.y, p->real.vel.z};
  orc_reset_agents(p, pos, vel, obstacles);
  orc_rollout_omp(p, n_threads);
  return best;
}

/* CfAgent::getPathLength, B/src/cf_agent.cpp:26-32 */
static double path_length(const agent_t *a) {
  double len = 0;
  for (int i = 0; i + 1 < a->n_pos; i++) len += vnorm(vsub(a->pos[i + 1], a->pos[i]));
  return len;
}

/* CfManager::evaluateAgents, B/src/cf_manager.cpp:293-356 */
int orc_evaluate(orc_planner *p, double k_goal_dist, double k_path_len,
                 double k_safe_dist, double k_workspace, const double *ws) {
  for (int i = 0; i < p->n_agents; i++) {
    const agent_t *a = &p->agents[i];
    double cost = 0;
    for (int k = 0; k < a->n_pos; k++) {
      v3 q = a->pos[k];
      double t;
      if (q.x > ws[0]) { t = fabs(q.x - ws[0]) * k_workspace; cost += t * t; }
      else if (q.x < ws[1]) { t = fabs(q.x - ws[1]) * k_workspace; cost += t * t; }
      if (q.y > ws[2]) { t = fabs(q.y - ws[2]) * k_workspace; cost += t * t; }
      else if (q.y < ws[3]) { t = fabs(q.y - ws[3]) * k_workspace; cost += t * t; }
      if (q.z > ws[4]) { t = fabs(q.z - ws[4]) * k_workspace; cost += t * t; }
      else if (q.z < ws[5]) { t = fabs(q.z - ws[5]) * k_workspace; cost += t * t; }
    }
    double goal_dist = dist_goal(a);
    if (goal_dist > p->approach) cost += goal_dist * k_goal_dist;
    cost += path_length(a) * k_path_len;
    cost += k_safe_dist / a->min_obs_dist;
    if (a->min_obs_dist < 2e-5) cost += 10000.0;
    p->costs[i] = cost;
  }
  int min_idx = 0;
  double min_cost = 1.7976931348623157e308; /* numeric_limits<double>::max() */
  for (int i = 0; i < p->n_agents; i++) {
    if (p->costs[i] < min_cost) { min_cost = p->costs[i]; min_idx = i; }
  }
  int take = 0;
  if (p->has_best) {
    if (p->costs[min_idx] < 0.9 * p->costs[p->best_id - 1]) take = 1;
    else min_idx = p->best_id - 1;
  } else {
    take = 1;
  }
  if (take) {                         /* best_agent_ = makeCopy() */
    p->has_best = 1;
    p->best_id = p->agents[min_idx].id;
    p->best_type = p->agents[min_idx].type;
    memcpy(p->best_rand, p->agents[min_idx].rand_vecs, sizeof(v3) * (size_t)p->n_obs);
  }
  return min_idx;
}

/* CfManager::moveRealEEAgent -> RealCfAgent::cfPlanner,
 * B/src/cf_manager.cpp:257-263, B/src/cf_agent.cpp:343-366 */
void orc_move_real(orc_planner *p, const double *obstacles, double dt,
                   int steps, int agent_id) {
  obs_t *obs = (obs_t *)malloc(sizeof(obs_t) * (size_t)p->n_obs);
  unpack_obstacles(obstacles, p->n_obs, obs);
  agent_t *a = &p->real;
  double k_attr = p->k_attr[agent_id], k_circ = p->k_circ[agent_id];
  double k_repel = p->k_repel[agent_id], k_damp = p->k_damp[agent_id];
  for (int s = 0; s < steps; s++) {
    a->force = V(0.0, 0.0, 0.0);
    double k_goal_scale = 1.0;
    if (gate_open(a)) {
      circ_force(a, obs, p->n_obs, k_circ, 0, p->best_type, p->best_rand);
      if (vnorm(a->force) > 1e-5) k_goal_scale = attractor_force_scaling(a, obs, p->n_obs);
    }
    repel_force(a, obs, p->n_obs, k_repel);
    attractor_force(a, k_attr, k_damp, k_goal_scale);
    update_pos_vel(a, dt);
  }
  free(obs);
}

/* CfAgent::setVelocity, B/src/cf_agent.cpp:54-61 */
static void set_velocity(agent_t *a, v3 vel) {
  double n = vnorm(vel);
  if (n > a->vel_max) a->vel = vscale(a->vel_max / n, vel);
  else a->vel = vel;
}

/* CfManager::resetEEAgents, B/src/cf_manager.cpp:246-255;
 * CfAgent::setObstacles, B/src/cf_agent.cpp:63-70 (radius is not copied) */
void orc_reset_agents(orc_planner *p, const double *pos, const double *vel,
                      const double *obstacles) {
  v3 q = V(pos[0], pos[1], pos[2]);
  v3 w = V(vel[0], vel[1], vel[2]);
  for (int i = 0; i < p->n_agents; i++) {
    agent_t *a = &p->agents[i];
    a->n_pos = 0;
    push_pos(a, q);
    set_velocity(a, w);
    for (int k = 0; k < p->n_obs; k++) {
      a->obstacles[k].pos = V(obstacles[7 * k + 0], obstacles[7 * k + 1], obstacles[7 * k + 2]);
      a->obstacles[k].vel = V(obstacles[7 * k + 3], obstacles[7 * k + 4], obstacles[7 * k + 5]);
      a->known[k] = p->real.known[k];
    }
    a->min_obs_dist = a->shell;
  }
}

/* ---- synchronous stepping API (no callers in the reference) ---- */

/* CfAgent::cfPlanner, B/src/cf_agent.cpp:278-300: `steps` steps with the CALLER's obstacle list (positions,
 * velocities AND radii; no obstacle advance), no loop guard, the agent's own known flags / rotation vectors, and
 * min_obs_dist_ tracking as in cfPrediction */
static void plan_steps(orc_planner *p, int i, const obs_t *obs, double dt, int steps) {
  agent_t *a = &p->agents[i];
  for (int s = 0; s < steps; s++) {
    a->force = V(0.0, 0.0, 0.0);
    double k_goal_scale = 1.0;
    if (gate_open(a)) {
      circ_force(a, obs, p->n_obs, p->k_circ[i], 1, a->type, a->rand_vecs);
      if (vnorm(a->force) > 1e-5) k_goal_scale = attractor_force_scaling(a, obs, p->n_obs);
    }
    repel_force(a, obs, p->n_obs, p->k_repel[i]);
    attractor_force(a, p->k_attr[i], p->k_damp[i], k_goal_scale);
    update_pos_vel(a, dt);
  }
}

/* CfManager::moveAgents / moveAgentsPar, B/src/cf_manager.cpp:274-291 */
void orc_move_agents(orc_planner *p, const double *obstacles, double dt, int steps) {
  obs_t *obs = (obs_t *)malloc(sizeof(obs_t) * (size_t)p->n_obs);
  unpack_obstacles(obstacles, p->n_obs, obs);
  for (int i = 0; i < p->n_agents; i++) plan_steps(p, i, obs, dt, steps);
  free(obs);
}

/* CfManager::moveAgent, B/src/cf_manager.cpp:265-272: cfPlanner(steps) repeated while the agent is farther than
 * 0.05 from the goal (run_prediction_ taken as true); at most max_calls calls (the reference has no bound).
 * Returns the number of cfPlanner calls made. */
int orc_move_agent(orc_planner *p, const double *obstacles, double dt, int steps, int id, int max_calls) {
  obs_t *obs = (obs_t *)malloc(sizeof(obs_t) * (size_t)p->n_obs);
  unpack_obstacles(obstacles, p->n_obs, obs);
  int calls = 0;
  while (dist_goal(&p->agents[id]) > 0.05 && calls < max_calls) {
    plan_steps(p, id, obs, dt, steps);
    calls++;
  }
  free(obs);
  return calls;
}

/* CfManager::setEEAgentPositions, B/src/cf_manager.cpp:220-224 */
void orc_set_agent_positions(orc_planner *p, const double *pos) {
  for (int i = 0; i < p->n_agents; i++) {
    p->agents[i].n_pos = 0;
    push_pos(&p->agents[i], V(pos[0], pos[1], pos[2]));
  }
}

/* CfManager::setEEAgentPosAndVels, B/src/cf_manager.cpp:238-244 */
void orc_set_agent_pos_and_vels(orc_planner *p, const double *pos, const double *vel) {
  for (int i = 0; i < p->n_agents; i++) {
    p->agents[i].n_pos = 0;
    push_pos(&p->agents[i], V(pos[0], pos[1], pos[2]));
    set_velocity(&p->agents[i], V(vel[0], vel[1], vel[2]));
  }
}

/* CfAgent::evalObstacleDistance, B/src/cf_agent.cpp:146-157, for every agent: min over ALL obstacles (the trailing
 * one included) of the unclamped surface distance, starting from the shell radius */
void orc_eval_obstacle_distance(const orc_planner *p, const double *obstacles, double *out) {
  for (int i = 0; i < p->n_agents; i++) {
    const agent_t *a = &p->agents[i];
    double min_dist = a->shell;
    for (int k = 0; k < p->n_obs; k++) {
      v3 o = V(obstacles[7 * k + 0], obstacles[7 * k + 1], obstacles[7 * k + 2]);
      double d = vnorm(vsub(latest(a), o)) - (a->rad + obstacles[7 * k + 6]);
      if (min_dist > d) min_dist = d;
    }
    out[i] = min_dist;
  }
}

/* install a best-agent copy (hysteresis reference / the real agent's heuristic) from outside: what survives
 * CfManager::init in the reference, and what an agent-range shard receives from the winner's owner */
void orc_set_best(orc_planner *p, int id, int type, const double *rand_vecs) {
  p->has_best = id > 0;
  p->best_id = id;
  p->best_type = type;
  if (rand_vecs) memcpy(p->best_rand, rand_vecs, sizeof(v3) * (size_t)p->n_obs);
}

int orc_tick(orc_planner *p, const double *obstacles, double dt,
             const double *cost_gains, const double *ws) {
  int best = orc_evaluate(p, cost_gains[0], cost_gains[1], cost_gains[2], cost_gains[3], ws);
  orc_move_real(p, obstacles, dt, 1, best);
  v3 np = latest(&p->real);
  double pos[3] = {np.x, np.y, np.z};
  double vel[3] = {p->real.vel.x, p->real.vel.y, p->real.vel.z};
  orc_reset_agents(p, pos, vel, obstacles);
  orc_rollout(p);
  return best;
}

/* CfManager::getLinkForce / CfAgent::bodyForce, B/src/cf_manager.cpp:169-182,
 * B/src/cf_agent.cpp:229-234. The force agents are GoalHeuristic agents built
 * in init (cf_manager.cpp:105-111) whose only used state is position, radius
 * and shell. */
void orc_link_force(orc_planner *p, int n, const double *link_pos,
                    const double *k_r_force, const double *obstacles,
                    double *out) {
  obs_t *obs = (obs_t *)malloc(sizeof(obs_t) * (size_t)p->n_obs);
  unpack_obstacles(obstacles, p->n_obs, obs);
  for (int i = 0; i < n; i++) {
    agent_t a;
    memset(&a, 0, sizeof(a));
    v3 q = V(link_pos[3 * i], link_pos[3 * i + 1], link_pos[3 * i + 2]);
    a.pos = &q; a.n_pos = 1; a.cap = 1;
    a.g_pos = p->goal;
    a.shell = p->real.shell; a.rad = p->real.rad;
    a.force = V(0.0, 0.0, 0.0);
    repel_force(&a, obs, p->n_obs, k_r_force[i]);
    out[3 * i] = a.force.x; out[3 * i + 1] = a.force.y; out[3 * i + 2] = a.force.z;
  }
  free(obs);
}

/* ------------------------------------------------------------------ */
int orc_n_agents(const orc_planner *p) { return p->n_agents; }
int orc_n_obs(const orc_planner *p) { return p->n_obs; }
int orc_capacity(const orc_planner *p) { return p->cap; }

void orc_get_paths(const orc_planner *p, double *paths, int32_t *n_points) {
  for (int i = 0; i < p->n_agents; i++) {
    const agent_t *a = &p->agents[i];
    n_points[i] = a->n_pos;
    memcpy(paths + (size_t)i * (size_t)p->cap * 3, a->pos, sizeof(v3) * (size_t)a->n_pos);
  }
}
void orc_get_costs(const orc_planner *p, double *costs) {
  memcpy(costs, p->costs, sizeof(double) * (size_t)p->n_agents);
}
void orc_get_path_lengths(const orc_planner *p, double *out) {
  for (int i = 0; i < p->n_agents; i++) out[i] = path_length(&p->agents[i]);
}
void orc_get_min_obs_dist(const orc_planner *p, double *out) {
  for (int i = 0; i < p->n_agents; i++) out[i] = p->agents[i].min_obs_dist;
}
void orc_get_success(const orc_planner *p, int32_t *out) {
  for (int i = 0; i < p->n_agents; i++) out[i] = p->agents[i].reached_goal;
}
void orc_get_agent_vel(const orc_planner *p, double *out) {
  for (int i = 0; i < p->n_agents; i++) memcpy(out + 3 * i, &p->agents[i].vel, sizeof(v3));
}
void orc_get_rot_vecs(const orc_planner *p, double *out) {
  for (int i = 0; i < p->n_agents; i++)
    memcpy(out + (size_t)i * (size_t)p->n_obs * 3, p->agents[i].rot, sizeof(v3) * (size_t)p->n_obs);
}
void orc_get_known(const orc_planner *p, int32_t *out) {
  for (int i = 0; i < p->n_agents; i++)
    for (int k = 0; k < p->n_obs; k++) out[(size_t)i * (size_t)p->n_obs + k] = p->agents[i].known[k];
}
void orc_get_real_state(const orc_planner *p, double *pos, double *vel, double *force) {
  v3 q = latest(&p->real);
  if (pos) memcpy(pos, &q, sizeof(v3));
  if (vel) memcpy(vel, &p->real.vel, sizeof(v3));
  if (force) memcpy(force, &p->real.force, sizeof(v3));
}
void orc_get_real_known(const orc_planner *p, int32_t *known, double *rot) {
  for (int k = 0; k < p->n_obs; k++) {
    if (known) known[k] = p->real.known[k];
    if (rot) memcpy(rot + 3 * k, &p->real.rot[k], sizeof(v3));
  }
}
int orc_get_real_path(const orc_planner *p, double *out, int max_points) {
  int n = p->real.n_pos < max_points ? p->real.n_pos : max_points;
  if (out) memcpy(out, p->real.pos, sizeof(v3) * (size_t)n);
  return p->real.n_pos;
}
double orc_dist_from_goal(const orc_planner *p) { return vnorm(vsub(p->goal, latest(&p->real))); }
int orc_best_type(const orc_planner *p) { return p->has_best ? p->best_type : -1; }
int orc_best_id(const orc_planner *p) { return p->has_best ? p->best_id : 0; }
int64_t orc_agent_steps(const orc_planner *p) { return p->agent_steps; }


/* ------------------------------------------------------------------ */
/* Set-point consumer: the 3-vector half of the controller that receives the planner's
 * set-points -- TrajectoryBuffer (B/src/trajectory_buffer.cpp:13-64, size 1: B/src/costp_controller.cpp:25) and
 * CoSTPController::reset (:88-109), fillBuffer (:289-297), followTrajectory's trajectory logic (:299-340),
 * absolutePositionControl's speed ramp (:193-201), getInstantaneousGoal (:144-155), getCurrentNominalGoal
 * (:138-142). The joint-space control behind it (dqrobotics) is out of scope. */
struct orc_consumer {
  /* TrajectoryBuffer(1) */
  v3 buf[1];
  int size, head, tail, full;
  int ready;
  v3 lg, cg, current_ng, last_ng, next_ig, current_ig;
  double next_ng, v_act, v_goal, reserve, min_motion;
  long accepted, refused, n_nan, too_close, inconsistent, updates;
};

orc_consumer *orc_consumer_create(void) {
  orc_consumer *c = (orc_consumer *)calloc(1, sizeof(*c));
  c->size = 1;
  c->reserve = 0.1;        /* costp_controller.h:79 */
  c->min_motion = 2e-6;    /* costp_controller.cpp:301 */
  return c;
}
void orc_consumer_destroy(orc_consumer *c) { free(c); }

/* CoSTPController::reset, :88-109 */
void orc_consumer_reset(orc_consumer *c, const double *ee_pos) {
  c->ready = 1;
  c->head = c->tail; c->full = 0;                       /* tb_->clear() */
  c->buf[(c->tail + 0) % c->size] = V(ee_pos[0], ee_pos[1], ee_pos[2]);
  c->lg = c->buf[(c->tail + 0) % c->size];
  c->next_ig = c->lg; c->cg = c->lg; c->current_ng = c->lg; c->last_ng = c->lg;
  c->next_ng = 0;
  c->current_ig = c->lg;
  c->v_act = 0;
  c->v_goal = 0;
}
int orc_consumer_ready(const orc_consumer *c) { return c->ready; }

/* CoSTPController::fillBuffer, :289-297; TrajectoryBuffer::put, trajectory_buffer.cpp:45-53 */
int orc_consumer_fill(orc_consumer *c, const double *goal) {
  int ok = 1;
  if (c->full) { ok = 0; c->refused++; }
  else {
    c->buf[c->head] = V(goal[0], goal[1], goal[2]);
    c->head = (c->head + 1) % c->size;
    c->full = c->head == c->tail;
  }
  if (c->full) c->ready = 0;
  return ok;
}

/* one 1 kHz cycle: followTrajectory :299-340, absolutePositionControl :193-201, getInstantaneousGoal :144-155;
 * out = the instantaneous goal */
void orc_consumer_update(orc_consumer *c, double v_max, double *out) {
  c->updates++;
  if (c->next_ng >= 1 || (c->v_act == 0 && c->next_ng == 0)) {
    int got_point = 0;
    c->lg = c->cg;
    if (!(!c->full && c->head == c->tail)) {            /* !tb_->empty() */
      c->cg = c->buf[c->tail];                          /* tb_->get() */
      c->full = 0;
      c->tail = (c->tail + 1) % c->size;
      got_point = 1;
    } else {
      c->next_ng = 0;
      c->v_act = 0;
    }
    c->ready = 1;
    if (got_point) {
      c->accepted++;
      if (isnan(c->cg.x) || isnan(c->cg.y) || isnan(c->cg.z)) c->n_nan++;
      if (vnorm(vsub(c->cg, c->lg)) < 1e-6) {
        c->too_close++;
        c->cg.z += c->min_motion;
        c->min_motion = -c->min_motion;
      }
      const double v = v_max * (1 - c->reserve);
      c->v_goal = dmin(vnorm(vsub(c->cg, c->lg)) * 100, v);
      double acos_gamma = vdot(vnormalized(vsub(c->cg, c->lg)), vsub(c->current_ng, c->lg));
      double e = c->v_goal * 0.001;
      double l = vnorm(vsub(c->lg, c->current_ng));
      double radicand = (acos_gamma * acos_gamma + e * e) - l * l;
      double b;
      if (radicand >= 0) b = acos_gamma + sqrt(radicand);
      else { b = 0; c->inconsistent++; }
      c->next_ng = b / vnorm(vsub(c->cg, c->lg));
    }
  }
  if (c->v_goal > c->v_act) {
    c->v_act += 0.001 * 0.05;
    if (c->v_act > c->v_goal) c->v_act = c->v_goal;
  } else {
    c->v_act = c->v_goal;
  }
  /* getInstantaneousGoal */
  v3 current_ig = c->next_ig;
  if (vnorm(vsub(c->current_ng, current_ig)) < c->v_act * 0.001) {
    c->last_ng = c->current_ng;                          /* getCurrentNominalGoal */
    c->current_ng = vadd(c->lg, vscale(c->next_ng, vsub(c->cg, c->lg)));
    c->next_ng += c->v_goal * 0.001 / vnorm(vsub(c->cg, c->lg));
  }
  c->next_ig = vadd(current_ig, vscale(c->v_act * 0.001, vnormalized(vsub(c->current_ng, current_ig))));
  c->current_ig = vadd(vscale(0.9, c->current_ig), vscale(0.1, current_ig));
  if (out) { out[0] = c->current_ig.x; out[1] = c->current_ig.y; out[2] = c->current_ig.z; }
}

/* VrepController::targetPoseCallback, B/src/vrep_controller.cpp:100-115 (v_max = velocity / 0.9, :291-292) */
long orc_consumer_deliver(orc_consumer *c, const double *set_point, double velocity, long max_cycles) {
  orc_consumer_fill(c, set_point);
  long n = 0;
  while (n < max_cycles) {
    orc_consumer_update(c, velocity / 0.9, NULL);
    n++;
    if (c->ready) break;
  }
  return n;
}

/* state = {v_goal, v_act, next_ng, cg[3], lg[3], current_ng[3], current_ig[3]} (15), counters = {accepted, refused,
 * nan, too_close, inconsistent, updates} (6) */
void orc_consumer_state(const orc_consumer *c, double *state, long *counters) {
  if (state) {
    state[0] = c->v_goal; state[1] = c->v_act; state[2] = c->next_ng;
    memcpy(state + 3, &c->cg, sizeof(v3)); memcpy(state + 6, &c->lg, sizeof(v3));
    memcpy(state + 9, &c->current_ng, sizeof(v3)); memcpy(state + 12, &c->current_ig, sizeof(v3));
  }
  if (counters) {
    counters[0] = c->accepted; counters[1] = c->refused; counters[2] = c->n_nan;
    counters[3] = c->too_close; counters[4] = c->inconsistent; counters[5] = c->updates;
  }
}

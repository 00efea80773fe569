/*
 * pmaf_oracle.h -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement, in plain C, of the reference's predictive multi-agent
 * circular-field planner tick (bimanual_planning_ros: CfAgent / CfManager).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libpmaf_hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path, and its sources need Eigen3/dqrobotics/ROS headers that are
 * absent from this image, so no reference build exists to pin against (see
 * DESIGN.md "Oracle"). The restatement follows the reference line by line
 * (each function cites the file:line it follows, B/ =
 * /root/reference/src/bimanual_planning_ros/) and is additionally checked
 * against the probe numbers recorded in SURVEY.md section 8(c).
 *
 * HOW TO PIN IT (one command, on a machine that can build the reference:
 * Eigen3 + the dqrobotics packages installed; NOT possible in this repo's
 * build container, where the script exits 77 = skipped):
 *     bash oracle/pin/make_pin.sh /path/to/predictive-multi-agent-framework
 *     python -m pytest tests/test_reference_pin.py -q -s
 * The script compiles the reference's cf_agent.cpp + cf_manager.cpp unmodified
 * with oracle/pin/pin_harness.cpp (it drives CfManager through the planner
 * node's call sequence, Random vectors from the scenario file, every rollout
 * run to its guard) and writes tests/golden/ref_*.json; the test holds this
 * oracle to them BIT FOR BIT under both 3-vector dot-product associations
 * (PMAF_DOT_RIGHT_ASSOC, orc_eval_order) and names the product library
 * variant that matches the reference build (include/pmaf.h, pmaf_eval_order).
 * Once those files are committed this header and DESIGN.md drop "unpinned".
 *
 * Conventions: all arithmetic IEEE double, compiled with -ffp-contract=off.
 * obstacles are flat [n_obs][7] = px,py,pz,vx,vy,vz,r; the LAST obstacle
 * (index n_obs-1) is the repulsive-only one (B/src/cf_agent.cpp:159-181),
 * indices 0..n_obs-2 generate circular fields (B/src/cf_agent.cpp:75).
 */
#ifndef PMAF_ORACLE_H
#define PMAF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Agent types: values of CfAgent::Type, B/include/bimanual_planning_ros/cf_agent.h:59-68 */
enum {
  ORC_REAL_AGENT = 0,
  ORC_GOAL_HEURISTIC = 1,
  ORC_OBSTACLE_HEURISTIC = 2,
  ORC_GOAL_OBSTACLE_HEURISTIC = 3,
  ORC_VEL_HEURISTIC = 4,
  ORC_RANDOM_AGENT = 5,
  ORC_HAD_HEURISTIC = 6
};

typedef struct orc_planner orc_planner;

/* exp() used by attractorForceScaling (B/src/cf_agent.cpp:220): 0 = libm exp
 * (reference-faithful, default), 1 = pmaf_portable_exp (same function as the
 * HIP kernels; bit-reproducible on any IEEE platform). Process-global. */
void orc_set_exp_mode(int mode);
int orc_get_exp_mode(void);
double pmaf_portable_exp(double x);
/* y[i] = pmaf_portable_exp(x[i]) (which = 1) or the host libm's exp(x[i]) (which = 0) */
void pmaf_exp_array(int which, const double *x, double *y, long n);

/*
 * CfManager::init on a default-constructed manager
 * (B/src/cf_manager.cpp:41-124). scal = {dt (= prediction_freq_multiple *
 * delta_t), velocity_max, approach_dist, detect_shell_rad, agent_mass,
 * radius}. gains = [4][n_agents] rows k_attr,k_circ,k_repel,k_damp.
 * types may be NULL -> reference population layout (cf_manager.cpp:70-104):
 * Had, Goal, Obstacle, GoalObstacle, Vel, then Random. random_vecs =
 * [n_agents][n_obs][3] unit vectors (only Random agents' rows are read);
 * replaces std::random_device (B/src/helper_functions.cpp:7-13).
 * mgr_init_pos = CfManager::init_pos_ at the time of init (agents are
 * constructed at that position).
 */
orc_planner *orc_create(int n_agents, int n_obs, int max_prediction_steps,
                        const double *scal, const double *goal,
                        const double *mgr_init_pos, const double *obstacles,
                        const double *gains, const int32_t *types,
                        const double *random_vecs);
void orc_destroy(orc_planner *p);

/* CfManager::setInitialPosition, B/src/cf_manager.cpp:226-236 */
void orc_set_initial_position(orc_planner *p, const double *pos);
/* 3-vector dot-product association of this build (0 left, 1 right: -DPMAF_DOT_RIGHT_ASSOC), pmaf_oracle.c header */
int orc_eval_order(void);
/* CfManager::setRealEEAgentPosition, B/src/cf_manager.cpp:216-218 */
void orc_set_real_position(orc_planner *p, const double *pos);

/*
 * startPrediction + every agent's cfPrediction inner loop run to completion
 * (guard false) + stopPrediction. B/src/cf_agent.cpp:302-341.
 */
void orc_rollout(orc_planner *p);
/* As orc_rollout but only agents [a0,a1) -- used by the threaded CPU baseline. */
void orc_rollout_range(orc_planner *p, int a0, int a1);

/* orc_rollout on n_threads OpenMP threads; orc_tick with that rollout */
void orc_rollout_omp(orc_planner *p, int n_threads);
int orc_tick_omp(orc_planner *p, const double *obstacles, double dt, const double *cost_gains, const double *ws,
                 int n_threads);

/* CfManager::evaluateAgents, B/src/cf_manager.cpp:293-356. ws = [xmax,xmin,ymax,ymin,zmax,zmin] */
int orc_evaluate(orc_planner *p, double k_goal_dist, double k_path_len,
                 double k_safe_dist, double k_workspace, const double *ws);
/* CfManager::moveRealEEAgent, B/src/cf_manager.cpp:257-263 */
void orc_move_real(orc_planner *p, const double *obstacles, double dt,
                   int steps, int agent_id);
/* CfManager::resetEEAgents, B/src/cf_manager.cpp:246-255 */
void orc_reset_agents(orc_planner *p, const double *pos, const double *vel,
                      const double *obstacles);
/* Synchronous stepping API: CfManager::moveAgents (B/src/cf_manager.cpp:274-291) ->
 * CfAgent::cfPlanner (B/src/cf_agent.cpp:278-300), moveAgent (:265-272; bounded by max_calls, returns the number
 * of cfPlanner calls), setEEAgentPositions (:220-224), setEEAgentPosAndVels (:238-244), and
 * CfAgent::evalObstacleDistance (B/src/cf_agent.cpp:146-157) for every agent, out [N]. */
void orc_move_agents(orc_planner *p, const double *obstacles, double dt, int steps);
int orc_move_agent(orc_planner *p, const double *obstacles, double dt, int steps, int id, int max_calls);
void orc_set_agent_positions(orc_planner *p, const double *pos);
void orc_set_agent_pos_and_vels(orc_planner *p, const double *pos, const double *vel);
void orc_eval_obstacle_distance(const orc_planner *p, const double *obstacles, double *out);
/* install a best-agent copy: id 1-based (0 = none), type, rand_vecs [n_obs][3] or NULL */
void orc_set_best(orc_planner *p, int id, int type, const double *rand_vecs);

/* the planCallback sequence stop/evaluate/move/reset/start+complete,
 * B/src/panda_bimanual_control.cpp:336-352. Returns best index. */
int orc_tick(orc_planner *p, const double *obstacles, double dt,
             const double *cost_gains, const double *ws);

/* CfManager::getLinkForce -> CfAgent::bodyForce, B/src/cf_manager.cpp:169-182,
 * B/src/cf_agent.cpp:229-234. link_pos [n][3], k_r_force [n], out [n][3]. */
void orc_link_force(orc_planner *p, int n, const double *link_pos,
                    const double *k_r_force, const double *obstacles,
                    double *out);

/* getters (B/src/cf_manager.cpp:184-214, cf_manager.h:73-92) */
int orc_n_agents(const orc_planner *p);
int orc_n_obs(const orc_planner *p);
int orc_capacity(const orc_planner *p);
void orc_get_paths(const orc_planner *p, double *paths /*[N][cap][3]*/,
                   int32_t *n_points /*[N]*/);
void orc_get_costs(const orc_planner *p, double *costs /*[N]*/);
void orc_get_path_lengths(const orc_planner *p, double *out /*[N]*/);
void orc_get_min_obs_dist(const orc_planner *p, double *out /*[N]*/);
void orc_get_success(const orc_planner *p, int32_t *out /*[N]*/);
void orc_get_agent_vel(const orc_planner *p, double *out /*[N][3]*/);
void orc_get_rot_vecs(const orc_planner *p, double *out /*[N][n_obs][3]*/);
void orc_get_known(const orc_planner *p, int32_t *out /*[N][n_obs]*/);
void orc_get_real_state(const orc_planner *p, double *pos, double *vel,
                        double *force);
void orc_get_real_known(const orc_planner *p, int32_t *known /*[n_obs]*/,
                        double *rot /*[n_obs][3]*/);
int orc_get_real_path(const orc_planner *p, double *out, int max_points);
double orc_dist_from_goal(const orc_planner *p);
int orc_best_type(const orc_planner *p); /* -1 if no best agent yet */
int orc_best_id(const orc_planner *p);   /* 1-based agent ID, 0 if none */
/* total agent-steps executed by orc_rollout calls so far (for timing) */
int64_t orc_agent_steps(const orc_planner *p);

/* Set-point consumer: TrajectoryBuffer(1) (B/src/trajectory_buffer.cpp:13-64) + the
 * trajectory half of CoSTPController (B/src/costp_controller.cpp:88-109, 138-155, 193-201, 289-340) + the
 * hand-over loop of VrepController::targetPoseCallback (B/src/vrep_controller.cpp:100-115). */
typedef struct orc_consumer orc_consumer;
orc_consumer *orc_consumer_create(void);
void orc_consumer_destroy(orc_consumer *c);
void orc_consumer_reset(orc_consumer *c, const double *ee_pos);
int orc_consumer_ready(const orc_consumer *c);
int orc_consumer_fill(orc_consumer *c, const double *goal);           /* 0: the buffer refused the point */
void orc_consumer_update(orc_consumer *c, double v_max, double *out /*[3] instantaneous goal, may be NULL*/);
long orc_consumer_deliver(orc_consumer *c, const double *set_point, double velocity, long max_cycles);
void orc_consumer_state(const orc_consumer *c, double *state /*[15]*/, long *counters /*[6]*/);

#ifdef __cplusplus
}
#endif
#endif

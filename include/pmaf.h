/*
 * pmaf.h -- C-ABI of the MI355X-native predictive multi-agent circular-field
 * planner tick (libpmaf_hip.so, hand-written HIP kernels for gfx950).
 *
 * This is the drop-in boundary for ONE path of the reference
 * (riddhiman13/predictive-multi-agent-framework, package bimanual_planning_ros):
 * the per-tick forward rollout of N virtual agents over an H-step horizon,
 * scoring, best-agent selection and the real agent's single step. The
 * reference has no FFI for it -- ghostplanner::cfplanner::CfManager is a plain
 * C++ class compiled into the planner node -- so every entry point below cites
 * the CfManager / CfAgent member it replaces (B/ = src/bimanual_planning_ros/).
 * The C++ facade in include/bimanual_planning_ros/cf_manager.h re-creates the
 * reference's class surface on top of these calls.
 *
 * Conventions
 *  - all arithmetic is IEEE double; flat row-major arrays; plain pointers.
 *    Every operation of the default kernels is the correctly rounded one the
 *    reference's compiled code performs, in its order (pmaf_eval_order); exp()
 *    (B/src/cf_agent.cpp:220) is glibc >= 2.28's algorithm restated
 *    (csrc/pmaf_device.hpp: portable_exp): the bits of std::exp on x86-64
 *    glibc hosts with FMA.
 *  - a handle batches P independent populations ("scenes": one CfManager
 *    each); every array argument carries a leading [P] dimension. The
 *    reference's single-manager use is P = 1.
 *  - obstacles are [P][n_obstacles][7] = px,py,pz,vx,vy,vz,radius. The LAST
 *    obstacle of a population is the repulsive-only one
 *    (B/src/cf_agent.cpp:159-181, reference README.md:80); the others generate
 *    circular fields (B/src/cf_agent.cpp:75).
 *  - agent index is 0-based (as returned by CfManager::evaluateAgents), agent
 *    IDs inside the handle are 1-based like the reference.
 *  - every function returns PMAF_OK (0) or a negative pmaf_status; the message
 *    is available from pmaf_last_error(). Nothing throws across the ABI.
 *  - a handle is not thread-safe (neither is CfManager: all calls come from the
 *    single-threaded ROS spinner, B/src/panda_bimanual_control.cpp:475).
 *  - numeric range: every input value must be finite and either 0 or
 *    2^-100 <= |x| <= 2^100 (PMAF_ERR_INVALID otherwise).
 *  - there is NO CPU fallback: without a HIP device pmaf_create fails with
 *    PMAF_ERR_DEVICE.
 */
#ifndef PMAF_H
#define PMAF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMAF_ABI_VERSION 7   /* 7: pmaf_pick_lanes_per_agent / pmaf_estimate_rollout_us (the mapping rule as a pure function), pmaf_get_priority_slices;
                              * pmaf_move_real with steps = 0 leaves no trace; 6: pmaf_eval_order (build-time evaluation-order policy); pmaf_set_real_position no longer waits
                              * for the running rollout; 5: PMAF_FLAG_CONTRACTED, pmaf_get_health, the winner path in pinned
                              * memory, the tick's time limit */

typedef enum pmaf_status {
  PMAF_OK = 0,
  PMAF_ERR_INVALID = -1, /* bad argument (NULL, size <= 0, empty obstacle list ...) */
  PMAF_ERR_DEVICE = -2,  /* HIP runtime / no device */
  PMAF_ERR_STATE = -3,   /* call not valid in the handle's current state */
  PMAF_ERR_NOMEM = -4
} pmaf_status;

/* values of CfAgent::Type, B/include/bimanual_planning_ros/cf_agent.h:59-68 */
typedef enum pmaf_agent_type {
  PMAF_REAL_AGENT = 0,
  PMAF_GOAL_HEURISTIC = 1,
  PMAF_OBSTACLE_HEURISTIC = 2,
  PMAF_GOAL_OBSTACLE_HEURISTIC = 3,
  PMAF_VEL_HEURISTIC = 4,
  PMAF_RANDOM_AGENT = 5,
  PMAF_HAD_HEURISTIC = 6
} pmaf_agent_type;

typedef struct pmaf_planner pmaf_planner;

/* pmaf_params.flags */
/* Opt-in fast arithmetic in the wave-per-agent rollout kernel: divisions and
 * square roots by v_rcp_f64 / v_rsq_f64 + two Newton steps (1-2 ulp) instead of
 * the correctly rounded IEEE sequences. Default (flag clear) is the strict mode
 * whose results are bit-identical to the CPU restatement. */
#define PMAF_FLAG_FAST_MATH 1
/* Use the compiler's fully general IEEE division / sqrt expansions in the
 * rollout kernels instead of the default hand-expanded ones (same bits for
 * all operands within 2^+-250, which the validated input range guarantees;
 * ~13 % slower). See csrc/pmaf_device.hpp "arithmetic policy". */
#define PMAF_FLAG_IEEE_SEQUENCES 2
/* pmaf_tick sleeps (20 us quanta) between its polls of the mailbox's sequence
 * number instead of spinning: no host core is burnt while the previous rollout
 * still runs (batch drivers issuing ticks back to back), at the price of the
 * wake-up latency (set-point up to a quantum + scheduler latency later).
 * Default: spin (lowest set-point latency). */
#define PMAF_FLAG_BLOCKING_WAIT 4
/* Opt-in CONTRACTED arithmetic in the rollout kernels (wave-per-agent and group kernels; ABI 5): PMAF_FLAG_FAST_MATH's
 * reciprocal / reciprocal-square-root sequences AND fused multiply-adds wherever the step forms a * b + c (dot products,
 * cross products, the integrator's a + b * s), plus an unordered (tree) force sum. The step of these kernels is bound
 * by its instruction count, and the FMA is the one FP64 instruction that retires two operations. Results are NOT
 * bit-identical to the CPU restatement. What holds instead (tests/test_tolerance_gpu.py, >= 50 closed-loop ticks against
 * the oracle's libm mode): on WELL-CONDITIONED scenes (BASELINE C1-C4, the six shipped dual_arms_* task scenes) the same
 * best-agent sequence and the selected trajectory within the north star's 1e-5 m; on CHAOTIC scenes it does NOT hold --
 * one of BASELINE C5's eight scenes (3.2e-3 m) and the shipped sim_kobo_dyn_spheres1/2/3 tasks (0.2 m; spheres3 selects
 * another agent at tick 1), the list CONTRACTED_EXCEEDS of that test file. The real agent's step (the set-point that is
 * published) is always evaluated in strict arithmetic. Default: flag clear. */
#define PMAF_FLAG_CONTRACTED 8

/*
 * Arguments of CfManager::init (B/src/cf_manager.cpp:41-124,
 * B/include/bimanual_planning_ros/cf_manager.h:93-102) for P populations.
 */
typedef struct pmaf_params {
  int32_t abi_version;          /* PMAF_ABI_VERSION */
  int32_t n_populations;        /* P >= 1 */
  int32_t n_agents;             /* N = k_a_ee.size() per population */
  int32_t n_obstacles;          /* obstacles.size() = M+1 >= 1 per population */
  int32_t max_prediction_steps; /* path capacity in points (H+1) */
  int32_t device;               /* HIP device ordinal; -1 = current device */
  int32_t lanes_per_agent;      /* 0 = auto; else 1,2,4,8,16,32,64 */
  int32_t flags;                /* PMAF_FLAG_* bits, 0 = defaults */
  double dt;                    /* prediction_freq_multiple * delta_t */
  double velocity_max;
  double approach_dist;
  double detect_shell_rad;
  double agent_mass;            /* reference default 1.0 */
  double radius;                /* reference default 0.05 */
  const double *goal;           /* [P][3] */
  const double *init_pos;       /* [P][3] CfManager::init_pos_ when init runs (agents are built there); NULL = zeros */
  const double *obstacles;      /* [P][n_obstacles][7] */
  const double *k_attr;         /* [P][N] k_a_ee */
  const double *k_circ;         /* [P][N] k_c_ee */
  const double *k_repel;        /* [P][N] k_r_ee */
  const double *k_damp;         /* [P][N] k_d_ee */
  const int32_t *agent_types;   /* [N] or NULL = reference layout Had,Goal,Obstacle,GoalObstacle,Vel,Random... (cf_manager.cpp:70-104) */
  const double *random_vecs;    /* [P][N][n_obstacles][3] unit vectors for Random agents (replaces std::random_device, B/src/helper_functions.cpp:7-13); NULL = all zero */
} pmaf_params;

/* CfManager::init / ~CfManager */
int pmaf_create(const pmaf_params *params, pmaf_planner **out);
int pmaf_destroy(pmaf_planner *h);
const char *pmaf_last_error(void);
int pmaf_abi_version(void);
/* Evaluation-order policy this library was BUILT with (csrc/build.sh, PMAF_VARIANT). The reference's arithmetic is
 * Eigen's, and one detail of it depends on how Eigen was compiled: a fixed-size 3-vector dot product / squaredNorm
 * (every norm, normalisation and projection of B/src/cf_agent.cpp:72-611) sums
 *   (a0 b0 + a1 b1) + a2 b2   PMAF_EVAL_ORDER_DOT_LEFT   Eigen 3.3 with a double-precision packet type: x86-64 SSE2,
 *                             aarch64 NEON (Redux.h, LinearVectorizedTraversal + CompleteUnrolling: predux of the first
 *                             packet, then the remaining coefficient) -- a stock `catkin build`; the DEFAULT library;
 *   a0 b0 + (a1 b1 + a2 b2)   PMAF_EVAL_ORDER_DOT_RIGHT  its non-vectorised redux (redux_novec_unroller splits 3 as
 *                             1 + 2): -DEIGEN_DONT_VECTORIZE, 32-bit ARM ...; the `rassoc` variant library.
 * Both are IEEE-conformant; they differ in the last bit of some sums, and a long rollout through many obstacles can
 * amplify that (profiles/r4_oracle_conditioning.txt: 2.3e-5 m on the shipped dual_arms_static1 scene, decimetres on
 * the sim_kobo_dyn_spheres* tasks). Pick the library whose order matches the Eigen build it replaces; oracle/pin/
 * holds the recipe that tells which one that is. Kernels, the manager's real step and the host-side getters all
 * follow the one switch (-DPMAF_DOT_RIGHT_ASSOC), and so does the CPU oracle the parity suite compares with. */
#define PMAF_EVAL_ORDER_DOT_LEFT 0
#define PMAF_EVAL_ORDER_DOT_RIGHT 1
int pmaf_eval_order(void);

/* CfManager::setInitialPosition, B/src/cf_manager.cpp:226-236. pos [P][3] */
int pmaf_set_initial_position(pmaf_planner *h, const double *pos);
/* CfManager::setRealEEAgentPosition, B/src/cf_manager.cpp:216-218 -> RealCfAgent::setPosition = push_back
 * (B/src/cf_agent.cpp:44-46): the measured position becomes the real agent's latest one (closed loop: the node calls it
 * in front of every tick when open_loop is false, B/src/panda_bimanual_control.cpp:333-335). pos [P][3]. Returns at once
 * (ABI 6): the position is left in pinned memory and read by the manager kernel of the NEXT pmaf_tick / pmaf_evaluate /
 * pmaf_move_real / pmaf_reset_agents -- no wait for the running rollout, no copy command in front of the tick; the
 * real-agent getters (pmaf_get_real_state, pmaf_get_dist_from_goal, pmaf_get_real_path) show it immediately. */
int pmaf_set_real_position(pmaf_planner *h, const double *pos);

/* CfManager::startPrediction (cf_manager.h:57-61): asynchronous launch of the
 * agent x horizon rollout kernel on the handle's stream. Rollouts always run
 * to their guard (full horizon or goal reached, B/src/cf_agent.cpp:310-311). */
int pmaf_start(pmaf_planner *h);
/* CfManager::stopPrediction (B/src/cf_manager.cpp:126-140): stream sync. */
int pmaf_stop(pmaf_planner *h);

/* The three calls below are what the unchanged node issues per tick (B/src/panda_bimanual_control.cpp:337-351). Each is
 * ONE manager launch whose result the host takes from the mailbox in pinned memory (no stream synchronisation); obstacle
 * lists are handed over through the mapped pinned buffer -- not at all when bit-identical to the resident one -- and
 * agent indices / reset states by value in the kernel arguments for <= 4 populations: the five-call tick costs ~33 us at
 * the C++ boundary, pmaf_tick ~13 us (profiles/r6_facade_latency.txt). */
/* CfManager::evaluateAgents, B/src/cf_manager.cpp:293-356.
 * cost_gains = {k_goal_dist,k_path_len,k_safe_dist,k_workspace},
 * ws = {xmax,xmin,ymax,ymin,zmax,zmin}; best_idx [P] out. */
int pmaf_evaluate(pmaf_planner *h, const double *cost_gains, const double *ws,
                  int32_t *best_idx);
/* CfManager::moveRealEEAgent, B/src/cf_manager.cpp:257-263. obstacles
 * [P][n_obstacles][7] live obstacles; agent_id [P]. */
int pmaf_move_real(pmaf_planner *h, const double *obstacles, double dt,
                   int32_t steps, const int32_t *agent_id);
/* CfManager::resetEEAgents, B/src/cf_manager.cpp:246-255. pos, vel [P][3]. */
int pmaf_reset_agents(pmaf_planner *h, const double *pos, const double *vel,
                      const double *obstacles);
/*
 * The planCallback sequence stop -> evaluate -> moveRealEEAgent(1 step) ->
 * resetEEAgents(next pos, next vel) -> start
 * (B/src/panda_bimanual_control.cpp:336-352) as two back-to-back launches.
 * Returns once best_idx / next_pos / next_vel (each [P], [P][3], [P][3]; may be
 * NULL) are on the host; the new rollout keeps running asynchronously, like
 * the reference's prediction threads. obstacles may be NULL (= unchanged).
 * A list that differs from the previous one in a field obstacle (compared bit for
 * bit) and is at rest makes the next reset recompute the Obstacle / GoalObstacle
 * heuristics' closest-other table (populations of more than 60 field obstacles on
 * the wave-per-agent kernels; M distances per obstacle, once): passing the same
 * static list every tick, as the reference's node does, costs nothing.
 */
int pmaf_tick(pmaf_planner *h, const double *obstacles, double dt,
              const double *cost_gains, const double *ws, int32_t *best_idx,
              double *next_pos, double *next_vel);

/* ---- failure detection on the tick (ABI 5; SURVEY.md section 5: the reference has none for the planner -- its consumer
 * logs a NaN set-point, B/src/costp_controller.cpp:317-319) ----
 * Time limit: pmaf_tick waits for the manager kernel's result at most PMAF_TICK_TIMEOUT_S seconds (environment, read
 * at pmaf_create; default 5) and then fails with PMAF_ERR_DEVICE instead of spinning on a hung device; the outputs are
 * not written in that case. The two kernels of that tick are still queued or running then (they read the call's staging
 * buffers and write the mailbox later), so the handle refuses every further pmaf_tick with PMAF_ERR_STATE until
 * pmaf_stop() has drained the stream (or the handle is recreated). A failed pmaf_tick never counts its obstacle list as
 * handed over: the retry passes it again.
 * Health word of the last pmaf_tick / pmaf_evaluate / pmaf_move_real per population (bits below), written by the
 * manager kernel next to the set-point. pmaf_tick itself still returns PMAF_OK with the (NaN) set-point in its
 * outputs -- the reference's planner publishes it too -- the caller decides (the C++ facade's planTick throws). */
#define PMAF_HEALTH_SETPOINT_NAN 1   /* the real agent's next position / velocity is not finite */
#define PMAF_HEALTH_FORCE_NAN 2      /* the force on the real agent is not finite (e.g. the Had heuristic's rotation vector
                                        for an obstacle centre on the agent-goal line, B/src/cf_agent.cpp:599-611) */
#define PMAF_HEALTH_ACC_CLAMPED 4    /* updatePositionAndVelocity's |a| <= 13 clamp acted on the real agent's step */
#define PMAF_HEALTH_COST_NAN 8       /* no agent had a comparable cost (all NaN): index 0 was kept, as the reference does */
int pmaf_get_health(pmaf_planner *h, int32_t *bits /* [P] */);

/* ---- the selected trajectory of every tick on the host (ABI 5) ----
 * What the reference's planCallback hands out per tick (B/src/panda_bimanual_control.cpp:340-347: the predicted paths,
 * the best one marked) without the copy of ALL paths pmaf_view_paths makes: once enabled, the manager kernel of every
 * pmaf_tick / pmaf_evaluate writes the SELECTED agent's path -- as it was scored -- into mapped pinned host memory
 * right behind the set-point (24 B per path point; the rollout launched by the same tick starts behind it).
 * pmaf_view_winner_path waits for the path of the LAST tick / evaluate and returns pointers into that memory:
 * *paths [P][cap][3] (entries past n_points[p] are unspecified), *n_points [P], *agent [P] (0-based index); valid until
 * the next pmaf_tick / pmaf_evaluate. Any of the three may be NULL. */
int pmaf_enable_winner_path(pmaf_planner *h, int32_t enable);
int pmaf_view_winner_path(pmaf_planner *h, const double **paths, const int32_t **n_points, const int32_t **agent);
/* entry of pmaf_tick -> the winner path on the host (pmaf_view_winner_path's return), library clock, microseconds;
 * one sample per pmaf_view_winner_path call that followed a pmaf_tick, oldest first; clears the record */
int pmaf_get_winner_path_times_us(pmaf_planner *h, double *out, int32_t max_n, int32_t *n);

/* ---- synchronous stepping API of the class surface (no callers in the reference) ----
 * CfManager::moveAgents / moveAgentsPar (B/src/cf_manager.cpp:274-291) ->
 * CfAgent::cfPlanner (B/src/cf_agent.cpp:278-300): every agent takes `steps`
 * steps of length dt from its CURRENT state through the caller's obstacle list
 * [P][n_obstacles][7] (positions, velocities and radii as given; no obstacle
 * advance, no loop guard). Paths are bounded by max_prediction_steps here
 * (PMAF_ERR_STATE if a path would outgrow it; the reference's grow without
 * bound). After a stepping call pmaf_start needs a pmaf_reset_agents /
 * pmaf_set_* first (rollouts start from the population's reset state). */
int pmaf_move_agents(pmaf_planner *h, const double *obstacles, double dt, int32_t steps);
/* CfManager::moveAgent (B/src/cf_manager.cpp:265-272): agent agent_id[p] repeats
 * cfPlanner(steps) while it is farther than 0.05 from the goal -- at most
 * max_calls times and while the path buffer has room; calls [P] (may be NULL)
 * = cfPlanner calls made. */
int pmaf_move_agent(pmaf_planner *h, const double *obstacles, double dt, int32_t steps,
                    const int32_t *agent_id, int32_t max_calls, int32_t *calls);
/* CfManager::setEEAgentPositions (B/src/cf_manager.cpp:220-224): every agent's
 * path restarts at pos [P][3]; velocities stay (for the stepping calls that
 * follow). Deviation: like after pmaf_move_agent(s), pmaf_start then needs a
 * pmaf_reset_agents / pmaf_set_initial_position first (PMAF_ERR_STATE
 * otherwise) -- in the reference a startPrediction() here would continue with
 * each agent's OWN velocity, known flags and advanced obstacle copies
 * (CfAgent::setPosition only clears the path, B/src/cf_agent.cpp:39-42), which
 * a rollout launched from the population's reset state cannot reproduce. The
 * reference has no caller of this method. */
int pmaf_set_agent_positions(pmaf_planner *h, const double *pos);
/* CfManager::setEEAgentPosAndVels (B/src/cf_manager.cpp:238-244): pos, vel [P][3]
 * (velocity clamped to velocity_max, CfAgent::setVelocity). Same rule for a
 * following pmaf_start as pmaf_set_agent_positions. */
int pmaf_set_agent_pos_and_vels(pmaf_planner *h, const double *pos, const double *vel);
/* CfAgent::evalObstacleDistance (B/src/cf_agent.cpp:146-157) of every agent at
 * its latest position against obstacles [P][n_obstacles][7]; out [P][N]. */
int pmaf_eval_obstacle_distance(pmaf_planner *h, const double *obstacles, double *out);

/* CfManager::getLinkForce -> CfAgent::bodyForce, B/src/cf_manager.cpp:169-182,
 * B/src/cf_agent.cpp:229-234: repel-only force of population `pop`'s last
 * obstacle at n link points. link_pos [n][3], k_r_force [n], out [n][3]. */
int pmaf_link_force(pmaf_planner *h, int32_t pop, int32_t n,
                    const double *link_pos, const double *k_r_force,
                    const double *obstacles, double *out);

/* ---- getters ---- */
/* Getters of rollout results (paths, costs, lengths, distances, flags,
 * rotation vectors) wait for the running rollout; the real-agent getters
 * (pmaf_get_real_state, pmaf_get_dist_from_goal, pmaf_get_real_path) are served
 * from host copies and return at once, like the reference's. */
/* getPredictedPaths / getNumPredictionSteps: paths [P][N][cap][3], n_points [P][N] */
int pmaf_get_paths(pmaf_planner *h, double *paths, int32_t *n_points);
/* The same data without the copy: *paths [P][N][cap][3] (entries past a path's
 * end are zero) and *n_points [P][N] point into the handle's pinned host mirror,
 * refreshed by ONE device-to-host copy per rollout generation however often the
 * getters are called (the reference's node calls getPredictedPaths() 3 x N
 * times per tick, B/src/panda_bimanual_control.cpp:341-344). Valid until the
 * next call that changes the predicted paths (start / tick / reset / set_* /
 * move_agents / load_state / attach). Either pointer may be NULL. */
int pmaf_view_paths(pmaf_planner *h, const double **paths, const int32_t **n_points);
/* costs of the last pmaf_evaluate / pmaf_tick, [P][N] */
int pmaf_get_costs(pmaf_planner *h, double *costs);
/* getPredictedPathLengths, B/src/cf_manager.cpp:192-198, [P][N] */
int pmaf_get_path_lengths(pmaf_planner *h, double *out);
/* CfAgent::getMinObsDist, [P][N] */
int pmaf_get_min_obs_dist(pmaf_planner *h, double *out);
/* getAgentSuccess, B/src/cf_manager.cpp:208-214, [P][N] (0/1) */
int pmaf_get_success(pmaf_planner *h, int32_t *out);
/* CfAgent::getVelocity of every predicted agent, [P][N][3] */
int pmaf_get_agent_velocities(pmaf_planner *h, double *out);
/* field_rotation_vecs_ [P][N][n_obstacles][3] and known_obstacles_ [P][N][n_obstacles] */
int pmaf_get_rotation_vectors(pmaf_planner *h, double *rot, int32_t *known);
/* getNextPosition / getNextVelocity / getEEForce (cf_manager.h:74-79), each [P][3], may be NULL */
int pmaf_get_real_state(pmaf_planner *h, double *pos, double *vel, double *force);
/* RealCfAgent known_obstacles_ [P][n_obstacles], field_rotation_vecs_ [P][n_obstacles][3] */
int pmaf_get_real_known(pmaf_planner *h, int32_t *known, double *rot);
/* getPlannedTrajectory (cf_manager.h:90-92) of population pop: copies up to
 * max_points points into out [max_points][3]; *n_total = path size. */
int pmaf_get_real_path(pmaf_planner *h, int32_t pop, double *out,
                       int32_t max_points, int32_t *n_total);
/* getDistFromGoal (cf_manager.h:87-89), [P] */
int pmaf_get_dist_from_goal(pmaf_planner *h, double *out);
/* getBestAgentType (cf_manager.h:73): type [P] (-1 = none yet), id [P] 1-based (0 = none) */
int pmaf_get_best(pmaf_planner *h, int32_t *type, int32_t *id);
/* getPredictionTimes (B/src/cf_manager.cpp:200-206; CfAgent::prediction_time_,
 * B/src/cf_agent.cpp:308-331): duration of each agent's last rollout in ns,
 * [P][N], from the device's constant-rate clock (wall_clock64) read by the
 * agent's wave at the start and the end of its rollout. */
int pmaf_get_prediction_times_ns(pmaf_planner *h, double *out);

/* ---- state transfer / sharding support (no reference equivalent) ---- */
/* restore hysteresis reference (best_agent_ survives CfManager::init): id
 * 1-based [P] (0 = none), type [P], rand [P][n_obstacles][3] */
int pmaf_set_best(pmaf_planner *h, const int32_t *id, const int32_t *type,
                  const double *rand_vecs);
/*
 * Winner record of a population = the result of its last selection
 * (pmaf_evaluate / pmaf_tick): PMAF_WINNER_RECORD_HEADER doubles
 *   {cost, agent index, n_points, agent type, real agent's position[3] (the
 *    set-point just published), its distance from the goal}
 * followed by the selected agent's predicted path[cap][3] (the path that was
 * scored; entries past n_points are zero) = (8 + 3*cap) doubles.
 * pmaf_write_winner_records packs the records of all P populations into DEVICE
 * memory `dst` on the handle's stream (call it after pmaf_evaluate, before
 * the agents are reset; pmaf_stop() or a stream-ordered consumer makes it
 * visible).
 */
#define PMAF_WINNER_RECORD_HEADER 8
int pmaf_write_winner_records(pmaf_planner *h, void *dst_device, size_t bytes);
size_t pmaf_winner_record_doubles(const pmaf_planner *h);
/* the hipStream_t the handle launches on (as void*), for stream-ordered consumers */
void *pmaf_stream(pmaf_planner *h);

/* ---- multi-GPU: one process per GPU, populations sharded over ranks ----
 * The path shards by POPULATION: every rank plans its own populations with its
 * own handle and there is no collective on the rollout's data path. Where a
 * run needs every population's winning trajectory on every rank (the dual-arm
 * coupling of BASELINE config 4, a goal sweep's global pick) the fixed-size
 * winner records are exchanged with ONE all-gather per tick -- RCCL
 * (ncclAllGather over xGMI) between GPUs. No reference equivalent: the
 * reference is a single-process CPU planner. */
typedef struct pmaf_comm pmaf_comm;
#define PMAF_COMM_ID_BYTES 128
/* ncclGetUniqueId: call on ONE rank and hand the 128 bytes to every rank
 * (MPI, a file, a socket, torch.distributed ...). */
int pmaf_comm_unique_id(void *id_out);
/* ncclCommInitRank on HIP device `device` (-1 = current device); collective
 * over all `world` ranks. */
int pmaf_comm_init_rccl(int32_t world, int32_t rank, const void *id, int32_t device, pmaf_comm **out);
/* wrap an ncclComm_t the caller already owns (not destroyed by
 * pmaf_comm_destroy), e.g. the communicator of a host application */
int pmaf_comm_from_rccl(void *nccl_comm, int32_t device, pmaf_comm **out);
/* Host-transport communicator (MPI, gloo, tests): `fn` must all-gather
 * bytes_per_rank bytes of HOST memory of every rank into recv (rank-major)
 * and return 0. Records are staged through pinned host memory. */
typedef int (*pmaf_host_allgather_fn)(void *ctx, const void *send, void *recv, size_t bytes_per_rank);
int pmaf_comm_init_host(int32_t world, int32_t rank, pmaf_host_allgather_fn fn, void *ctx, pmaf_comm **out);
int pmaf_comm_destroy(pmaf_comm *c);
int pmaf_comm_world(const pmaf_comm *c);
int pmaf_comm_rank(const pmaf_comm *c);
/* blocking all-gather of n_per_rank doubles of HOST data per rank (small
 * control-plane exchanges: agent-range cost vectors, set-points). */
int pmaf_comm_allgather(pmaf_comm *c, const double *send, double *recv, size_t n_per_rank);
/* CfManager::evaluateAgents' selection rule (B/src/cf_manager.cpp:336-353) on
 * a cost vector gathered from agent-range shards: first minimum, then the 0.9
 * hysteresis against prev_best (global 0-based index, -1 = none). */
int32_t pmaf_select_best(const double *costs, int32_t n, int32_t prev_best);

/* One-shot exchange: pmaf_write_winner_records into an internal send buffer,
 * then the all-gather of all ranks' P records into DEVICE memory recv_device
 * [world][P][record] -- with an RCCL communicator both are enqueued on the
 * handle's stream with no host synchronisation in between (pmaf_stop() or a
 * stream-ordered consumer makes the result visible); a host communicator
 * blocks. Same call-time rule as pmaf_write_winner_records. */
int pmaf_allgather_winners(pmaf_planner *h, pmaf_comm *c, void *recv_device, size_t bytes);
/* Per-tick exchange for the fused tick: once a communicator is attached,
 * every pmaf_tick / pmaf_evaluate publishes the winner records of its
 * selection and all-gathers them on a second stream while the next rollout
 * runs (the handle then keeps two path buffers so the rollout cannot overwrite
 * the path being sent). c = NULL detaches. The communicator must outlive the
 * attachment. */
int pmaf_attach_comm(pmaf_planner *h, pmaf_comm *c);
/* (The handle keeps two exchange slots: a tick never waits for the collective
 * of the tick before it, only -- when it comes to reuse that slot -- for the
 * one two ticks back.)
 * wait for the exchange of the last pmaf_tick / pmaf_evaluate; *records =
 * [world][P][record] doubles in pinned host memory, valid until the next
 * pmaf_tick / pmaf_evaluate; *n_doubles = world * P * record. */
int pmaf_winners_wait(pmaf_planner *h, const double **records, size_t *n_doubles);
/* device copy of the same table (valid after pmaf_winners_wait) */
void *pmaf_winners_device(pmaf_planner *h);
/* duration of the completed exchanges' all-gathers in microseconds (device
 * time between the events around ncclAllGather, i.e. including the wait for
 * the slowest rank; host communicator: wall time of the callback), oldest
 * first, at most max_n; *n = number written. Clears the record. */
int pmaf_get_exchange_times_us(pmaf_planner *h, double *out, int32_t max_n, int32_t *n);
/* host-side clock of the newest pmaf_tick calls (at most 8192 are kept), oldest first, in microseconds from the call's
 * entry: enqueue_us = both launches (k_manager, rollout) handed to the stream, setpoint_us = best index + next
 * set-point on the host (what the reference's planCallback publishes, B/src/panda_bimanual_control.cpp:329-369;
 * the rollout keeps running behind it). Either array may be NULL; *n = number written; clears the record.
 * Measured inside the library, so a caller in an interpreted language sees the path's latency, not its own. */
int pmaf_get_tick_times_us(pmaf_planner *h, double *enqueue_us, double *setpoint_us, int32_t max_n, int32_t *n);

/* ---- peer mailboxes: header-only exchange WITHOUT a collective (ABI 3) ----
 * Where a population needs only another population's set-point of the previous
 * tick (BASELINE config 4: each arm's trailing repulsive obstacle is the other
 * arm's end effector, B/src/cf_agent.cpp:159-181 for the obstacle's role), a
 * collective per tick puts a launch + rendezvous on the control path. Instead
 * every rank owns an INBOX in its device memory that is mapped into every peer
 * process (hipIpcGetMemHandle / hipIpcOpenMemHandle; peer GPUs reach it over
 * xGMI). The manager kernel of pmaf_tick number t
 *   - stores the 8-double record header of each of its populations, then the
 *     sequence number t behind a system-scope release, straight into slot
 *     [t & 1][rank][pop] of EVERY rank's inbox, and
 *   - for a population coupled with pmaf_peer_couple, reads the header with
 *     sequence number t-1 of (src_rank, src_pop) from its OWN inbox (local HBM)
 *     and uses its position[3] as the live trailing obstacle (velocity 0, the
 *     given radius) of this tick's real step and of the rollout it starts.
 * No host, no stream, no collective is involved; the winner-record all-gather
 * above stays what it was (the path table, off the control path). Every rank
 * must issue the same number of pmaf_tick calls; the other entry points
 * (pmaf_evaluate ...) neither publish nor consume.
 * TOPOLOGY: couplings must be PAIRWISE MUTUAL (population a reads b's header and b reads a's -- the dual-arm case).
 * The two parity slots per source rest on it: a source cannot publish tick t+1 (which overwrites header t-1) before it
 * has read the consumer's header t. A one-way coupling or a ring of three or more has no such back-pressure; the
 * consumer detects it from the coupling the publisher writes into its header and pmaf_tick fails with PMAF_ERR_STATE
 * (likewise when a source is found to have run ahead). Not part of a checkpoint
 * (reconnect after pmaf_load_state). No reference equivalent. */
#define PMAF_PEER_HANDLE_BYTES 128
/* allocate this handle's inbox for `world` ranks and export it: hand the
 * PMAF_PEER_HANDLE_BYTES bytes of every rank to every rank (any transport) */
int pmaf_peer_export(pmaf_planner *h, int32_t world, void *handle_out);
/* handles = [world][PMAF_PEER_HANDLE_BYTES] in rank order (handles[rank] = this
 * handle's own export). Handles exported by the same process are used directly
 * (several handles of one process: one host driving several GPUs, tests), the
 * others are opened with hipIpcOpenMemHandle. Every rank must hold the same
 * number of populations. */
int pmaf_peer_connect(pmaf_planner *h, int32_t world, int32_t rank, const void *handles);
/* couple population `pop`'s trailing obstacle to the set-point of population
 * src_pop on rank src_rank (src_rank < 0: uncouple). init_pos [3] (may be
 * NULL) = the source's position the FIRST tick uses (only before the first
 * pmaf_tick after pmaf_peer_connect); without it the first tick waits for a
 * header that only a previous tick could have published. */
int pmaf_peer_couple(pmaf_planner *h, int32_t pop, int32_t src_rank, int32_t src_pop, double radius,
                     const double *init_pos);
/* every rank must have stopped ticking (barrier) before ANY rank disconnects or
 * destroys its handle: peers store into this rank's inbox */
int pmaf_peer_disconnect(pmaf_planner *h);
/* observability / tests: the newest header [world][P][8] and its sequence
 * number [world][P] (-1: none yet) in this rank's inbox; a small blocking copy */
int pmaf_peer_read(pmaf_planner *h, double *headers, double *seq);
/* per pmaf_tick since the last call (oldest first, at most max_n; clears the
 * record): wait_us = time the manager kernel waited for the coupled header
 * (what the control path pays for the coupling; ~0 when the ranks keep pace),
 * publish_us = its stores into the peers' inboxes incl. the system-scope fence
 * (device clock) */
int pmaf_get_peer_times_us(pmaf_planner *h, double *wait_us, double *publish_us, int32_t max_n, int32_t *n);
/* what the connection rests on: *fine_grained = 1 if this handle's inbox is fine-grained device memory (stores of
 * another GPU become visible to a running kernel), 0 if the runtime could only export plain device memory -- then
 * pmaf_peer_connect refuses peers on ANOTHER device (PMAF_ERR_DEVICE) and only same-device peers (several processes on
 * one GPU) can be connected. Either pointer may be NULL. */
int pmaf_peer_info(pmaf_planner *h, int32_t *fine_grained, int32_t *world);
/* hipGetDeviceCount (0 without a usable HIP device) */
int pmaf_device_count(void);

/* ---- checkpoint / resume (no reference equivalent: its state lives in RAM) ---- */
/* Serialise the complete planner state of a handle (agents' rotation vectors
 * and known flags, paths, real agent incl. its trajectory, best-agent copy,
 * obstacle tables, scoring parameters) into a caller buffer of at least
 * pmaf_state_size(h) bytes, and restore it into a handle created with the same
 * dimensions. After a restore the planner continues bit-identically. */
size_t pmaf_state_size(const pmaf_planner *h);
int pmaf_save_state(pmaf_planner *h, void *blob, size_t bytes);
int pmaf_load_state(pmaf_planner *h, const void *blob, size_t bytes);

/* ---- measurement ---- */
/* HIP-event timing of the rollout launches: 0 off, 1 every launch, n > 1 every n-th launch (the events ride on the
 * kernel's own dispatch packet, but a timed dispatch still costs 3-5 us of device time per tick -- tools/ticklat.py,
 * profiles/r4_tick_overhead.txt -- so a throughput measurement samples) */
int pmaf_set_profiling(pmaf_planner *h, int32_t enable);
/* rollout launches since the last pmaf_reset_kernel_stats, timed or not (pmaf_get_kernel_stats's `launches` counts the
 * timed ones while profiling is on) */
int pmaf_get_launch_count(pmaf_planner *h, int64_t *launches);
/* accumulated rollout-kernel time (ms), launches and agent-steps since the last reset_stats */
int pmaf_get_kernel_stats(pmaf_planner *h, double *rollout_ms, int64_t *launches,
                          int64_t *agent_steps);
int pmaf_reset_kernel_stats(pmaf_planner *h);
/* chosen lanes-per-agent and grid of the rollout kernel */
int pmaf_get_launch_config(pmaf_planner *h, int32_t *lanes_per_agent,
                           int32_t *n_blocks, int32_t *lds_bytes);
/* Wave-per-agent handles with 61..256 field obstacles: how many waves share an agent's rollout (csrc/pmaf_k_mw.hip:
 * one block of 2..4 waves per agent, <= 64 obstacles per wave, one LDS hand-off per step) and how many obstacles each
 * wave holds; waves_per_agent = 1: the one-wave kernels (2 / 4 obstacle slots per lane; always with
 * PMAF_FLAG_IEEE_SEQUENCES). Chosen at pmaf_create while
 * every wave of the launch gets a SIMD of its own; PMAF_MW=0 in the environment keeps the one-wave kernels, PMAF_MW=3|4
 * asks for more waves than the obstacle count needs (tests, timing). Results are bit-identical either way. */
int pmaf_get_waves_per_agent(pmaf_planner *h, int32_t *waves_per_agent, int32_t *obstacles_per_wave);
/* Whether the handle's rollout launches run the wave-per-agent kernel's PRIORITY-SLICING loop (k_rollout_w64_sliced): with
 * 1 025 ... 2 048 one-slot wave-per-agent rollouts in the handle two waves share a SIMD, the issue arbiter would serve the older
 * one first (the launch then lasts 1.64 x a lone wave's rollout), and the two trade issue priority in slices of the 100 MHz
 * wall clock instead (slice_ticks x 10 ns each, the younger wave holding `younger_of_8` of every eight) so that both finish
 * together: -4 ... -7 % per launch (profiles/r6_slice_sweep.txt). Scheduling only -- the arithmetic is the same instruction
 * sequence, results are bit-identical. No counterpart in the reference (the OS schedules its threads). */
int pmaf_get_priority_slices(pmaf_planner *h, int32_t *enabled, int32_t *slice_ticks, int32_t *younger_of_8);
/* The mapping rule as a pure function (no handle, no device): the lanes-per-agent mapping pmaf_create chooses for
 * n_populations x n_agents agents and n_field_obstacles circular-field obstacles (M, without the trailing repulsive one)
 * when pmaf_params.lanes_per_agent is 0, on a device with n_simds SIMDs (0 = MI355X's 1024). The reference has no
 * counterpart (it runs one std::thread per agent, B/src/cf_manager.cpp:118-123); this is the scheduling decision that
 * replaces it. 0 on invalid arguments. The rule is a table of measured launch times (csrc/pmaf_lpa_model.hpp,
 * profiles/r6_lpa_grid.txt); pmaf_estimate_rollout_us returns its estimate of the rollout kernel's duration in
 * microseconds for one mapping (lanes_per_agent in {64, 32, 16, 8}; horizon = steps per rollout) or a negative value
 * when the mapping is not offered for that obstacle count (a narrower mapping than the wave per agent holds at most two
 * obstacles per lane). Estimates, not promises: capacity planning (how many populations fit a control period) and the
 * tests that hold the rule to the measured grid (tests/test_lpa_model.py). */
int32_t pmaf_pick_lanes_per_agent(int32_t n_agents, int32_t n_populations, int32_t n_field_obstacles, int32_t n_simds);
double pmaf_estimate_rollout_us(int32_t lanes_per_agent, int32_t n_agents, int32_t n_populations, int32_t n_field_obstacles,
                                int32_t horizon, int32_t n_simds);

/* Measurement tooling (tools/slackprof): from now on the handle's rollout launches run `kernel_name` out of the code
 * object file at `code_object_path` (same arguments, grid and LDS as the built-in wave-per-agent kernel: the product
 * kernel's own assembly with delay instructions inserted); NULL path = back to the built-in kernels. Wave-per-agent
 * handles only. */
int pmaf_debug_external_rollout(pmaf_planner *h, const char *code_object_path, const char *kernel_name);
/* Fault injection for the tick's time limit (tests): while enabled, the manager kernel of pmaf_tick does not publish
 * its sequence number, so the host's wait can only end by the time limit (or by the stream running empty). */
int pmaf_debug_withhold_mailbox(pmaf_planner *h, int32_t enable);

/* Self-test of the device arithmetic the parity argument rests on: evaluates
 * op over n elements ON THE GPU (0: a/b, 1: sqrt(a), 2: the kernels' portable exp(a), 3: a*b,
 * 4: a+b; 5: the kernels' guarded sqrt, 6: guarded divide, 7 / 8: vector /
 * scalar through the guarded / the compiler's divide; 9 / 10: a / sqrt(b)
 * through the kernels' shared-reciprocal sequence / the compiler; 11 / 12: the
 * fixup-free a / b and a / sqrt(b) used where the divisor is a positive
 * normal). Tests compare the
 * results bitwise with the host's IEEE results. */
int pmaf_debug_math(int32_t op, int32_t n, const double *a, const double *b, double *out);

#ifdef __cplusplus
}
#endif
#endif /* PMAF_H */

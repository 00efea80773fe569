// cf_manager.h -- host-side mirror of the reference's planner surface:
// ghostplanner::cfplanner::CfManager (B/include/bimanual_planning_ros/cf_manager.h:18-137,
// B/ = reference src/bimanual_planning_ros/) re-created on top of the C-ABI of
// libpmaf_hip.so (include/pmaf.h). Same class name, namespace, method names,
// argument order, defaults and return types as the reference, so a caller
// written against the reference (B/src/panda_bimanual_control.cpp:329-369,
// 463-471, 501-518) compiles against this header unchanged.
//
// What differs, deliberately (see DESIGN.md "Deviations"):
//  * startPrediction() launches the agent x horizon rollout kernel
//    asynchronously on the GPU, stopPrediction() waits for it; rollouts always
//    run to their guard instead of being cut by wall clock
//    (B/src/cf_agent.cpp:310-311).
//  * Random agents draw their vectors from a seeded generator
//    (setRandomSeed) instead of std::random_device
//    (B/src/helper_functions.cpp:7-13).
//  * moveAgent / moveAgents / moveAgentsPar / setEEAgentPositions /
//    setEEAgentPosAndVels (no callers in the reference, B/src/cf_manager.cpp:220-291)
//    run on the device with fixed-size path buffers (max_prediction_steps points).
//  * the class is movable but not copyable (the reference declares its copy
//    operations defaulted, which the compiler deletes: unique_ptr / std::thread
//    members, cf_manager.h:19-21,34,52-55).
//  * getPredictedPaths() returns a const reference to a cached copy instead of a
//    new copy per call (source compatible with the node's uses).
//  * errors of the device layer surface as std::runtime_error.
//
// Vector type: Eigen::Vector3d when PMAF_USE_EIGEN is defined (a ROS box),
// otherwise the small ghostplanner::cfplanner::Vec3 below (same accessors).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bimanual_planning_ros/obstacle.h"
#include "pmaf.h"

namespace ghostplanner {
namespace cfplanner {

// Eigen::Matrix<double, 6, 1> stand-in for des_ws_limits when Eigen is absent
#ifdef PMAF_USE_EIGEN
using Vector6d = Eigen::Matrix<double, 6, 1>;
#else
struct Vector6d {
  std::array<double, 6> d{};
  double &operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
};
#endif

class CfManager {
  pmaf_planner *h_ = nullptr;
  int n_agents_ = 0;
  int n_obs_ = 0;
  int cap_ = 0;
  Vector3d init_pos_{0.0, 0.0, 0.0};
  Vector3d goal_pos_{0.0, 0.0, 0.0};
  std::vector<double> k_r_force_;
  std::vector<double> random_vecs_;       // [N][n_obs][3] of the current population
  // best_agent_ survives init() in the reference (cf_manager.h:20, cf_manager.cpp:41-124)
  int best_id_ = 0, best_type_ = -1;
  std::vector<double> best_rand_;         // [n_obs][3]
  uint64_t seed_ = 0x9E3779B97F4A7C15ull;
  int device_ = -1;
  // getPredictedPaths() is called 3 x N times per tick by the reference's node when
  // visualize_predicted_paths is on (B/src/panda_bimanual_control.cpp:341-344): the
  // converted paths are kept until a call that changes them
  std::vector<std::vector<Vector3d>> paths_cache_;
  bool paths_cached_ = false;
  pmaf_comm *comm_ = nullptr;             // attached communicator (not owned)
  // planTick: std::runtime_error on a non-finite set-point / force. OFF by default, like the reference, which publishes
  // the NaN (its consumer logs it, B/src/costp_controller.cpp:317-319): the device state has already advanced to NaN when
  // the throw comes, so every later tick throws too and an uncaught exception ends a 100 Hz node. Opt in with
  // setThrowOnNumericFault(true) and re-init() after catching; getHealth() reports the same bits without throwing.
  bool throw_on_numeric_fault_ = false;
  bool selected_path_on_ = false;         // enableSelectedPath survives a re-init
  void touch() { paths_cached_ = false; }

  static void check(int rc, const char *what) {
    if (rc != PMAF_OK) throw std::runtime_error(std::string(what) + ": " + pmaf_last_error());
  }
  static std::vector<double> flat(const std::vector<Obstacle> &obstacles) {
    std::vector<double> o(obstacles.size() * 7);
    for (size_t i = 0; i < obstacles.size(); ++i) {
      const Vector3d p = obstacles[i].getPosition(), v = obstacles[i].getVelocity();
      double *r = &o[i * 7];
      r[0] = p.x(); r[1] = p.y(); r[2] = p.z(); r[3] = v.x(); r[4] = v.y(); r[5] = v.z();
      r[6] = obstacles[i].getRadius();
    }
    return o;
  }
  void require() const {
    if (!h_) throw std::logic_error("CfManager: init() has not been called");
  }
  void remember_best() {
    if (!h_) return;
    int32_t type = -1, id = 0;
    if (pmaf_get_best(h_, &type, &id) == PMAF_OK && id > 0) {
      best_id_ = id;
      best_type_ = type;
      best_rand_.assign(random_vecs_.begin() + (size_t)(id - 1) * n_obs_ * 3,
                        random_vecs_.begin() + (size_t)id * n_obs_ * 3);
    }
  }

 public:
  CfManager() = default;
  CfManager(const Vector3d agent_pos, const Vector3d goal_pos, const double delta_t,
            const std::vector<Obstacle> &obstacles, const std::vector<double> &k_a_ee,
            const std::vector<double> &k_c_ee, const std::vector<double> &k_r_ee,
            const std::vector<double> &k_d_ee, const std::vector<double> &k_manip,
            const std::vector<double> &k_r_force, const double velocity_max = 0.5,
            const double approach_dist = 0.25, const double detect_shell_rad = 0.8,
            const size_t max_prediction_steps = 1500, const size_t prediction_freq_multiple = 1,
            const double agent_mass = 1.0, const double radius = 0.01) {
    init_pos_ = agent_pos;
    // the reference's constructor forwards only the first twelve arguments to
    // init (cf_manager.cpp:38-39): the remaining ones take init's defaults
    (void)max_prediction_steps; (void)prediction_freq_multiple; (void)agent_mass; (void)radius;
    init(goal_pos, delta_t, obstacles, k_a_ee, k_c_ee, k_r_ee, k_d_ee, k_manip, k_r_force, velocity_max,
         approach_dist, detect_shell_rad);
  }
  ~CfManager() { if (h_) pmaf_destroy(h_); }
  CfManager(const CfManager &) = delete;             // implicitly deleted in the reference too (see header comment)
  CfManager &operator=(const CfManager &) = delete;
  CfManager(CfManager &&o) noexcept { *this = std::move(o); }    // cf_manager.h:53
  CfManager &operator=(CfManager &&o) noexcept {                 // cf_manager.h:55
    if (this != &o) {
      if (h_) pmaf_destroy(h_);
      h_ = o.h_; o.h_ = nullptr;
      n_agents_ = o.n_agents_; n_obs_ = o.n_obs_; cap_ = o.cap_;
      init_pos_ = o.init_pos_; goal_pos_ = o.goal_pos_;
      k_r_force_ = std::move(o.k_r_force_);
      random_vecs_ = std::move(o.random_vecs_);
      best_id_ = o.best_id_; best_type_ = o.best_type_;
      best_rand_ = std::move(o.best_rand_);
      seed_ = o.seed_; device_ = o.device_; comm_ = o.comm_;
      paths_cache_ = std::move(o.paths_cache_);
      paths_cached_ = o.paths_cached_; o.paths_cached_ = false;
      random_vecs_override_ = std::move(o.random_vecs_override_);
    }
    return *this;
  }

  // ---- build-specific knobs (no reference equivalent) ----
  void setRandomSeed(uint64_t seed) { seed_ = seed; }
  void setDevice(int device) { device_ = device; }
  // explicit Random-agent vectors [N][n_obs][3] for the next init()
  void setRandomVectors(const std::vector<double> &v) { random_vecs_override_ = v; }
  pmaf_planner *handle() { return h_; }
  // Sharded runs (one process per GPU): from now on every evaluateAgents / planTick all-gathers this
  // manager's winner record with the other ranks' (ncclAllGather enqueued by the library beside the next rollout).
  // The communicator (pmaf_comm_init_rccl / pmaf_comm_from_rccl / pmaf_comm_init_host) must outlive the manager or be
  // detached with attachCommunicator(nullptr); a later init() keeps it attached.
  void attachCommunicator(pmaf_comm *comm) {
    comm_ = comm;
    if (h_) check(pmaf_attach_comm(h_, comm), "attachCommunicator");
  }
  // Winner records of ALL ranks after the last evaluateAgents / planTick: [world][8 + 3 max_prediction_steps] doubles
  // = cost, agent index, n_points, agent type, next set-point[3], goal distance, the winning path (zero-padded)
  std::vector<double> gatherWinners() {
    require();
    const double *t = nullptr;
    size_t n = 0;
    check(pmaf_winners_wait(h_, &t, &n), "gatherWinners");
    return std::vector<double>(t, t + n);
  }
  size_t winnerRecordDoubles() const { return h_ ? pmaf_winner_record_doubles(h_) : 0; }

  // CfManager::init, B/src/cf_manager.cpp:41-124
  void init(const Vector3d goal_pos, const double delta_t, const std::vector<Obstacle> &obstacles,
            const std::vector<double> &k_a_ee, const std::vector<double> &k_c_ee,
            const std::vector<double> &k_r_ee, const std::vector<double> &k_d_ee,
            const std::vector<double> &k_manip, const std::vector<double> &k_r_force,
            const double velocity_max = 0.5, const double approach_dist = 0.25,
            const double detect_shell_rad = 0.8, const size_t max_prediction_steps = 1500,
            const size_t prediction_freq_multiple = 1, const double agent_mass = 1.0,
            const double radius = 0.05) {
    if (!((k_a_ee.size() == k_c_ee.size()) && (k_c_ee.size() == k_r_ee.size()) &&
          (k_c_ee.size() == k_manip.size()) && (k_d_ee.size() == k_a_ee.size())))
      throw std::invalid_argument("CfManager::init: gain vectors must have equal sizes");  // assert, :50
    remember_best();
    touch();
    if (h_) { pmaf_destroy(h_); h_ = nullptr; }
    goal_pos_ = goal_pos;
    k_r_force_ = k_r_force;
    n_agents_ = (int)k_a_ee.size();
    n_obs_ = (int)obstacles.size();
    cap_ = (int)max_prediction_steps;
    // RandomCfAgent ctor (cf_agent.h:338-342): n_obs normalised U(-1,1)^3 vectors per agent
    if (random_vecs_override_.size() == (size_t)n_agents_ * n_obs_ * 3) {
      random_vecs_ = random_vecs_override_;
    } else {
      random_vecs_.assign((size_t)n_agents_ * n_obs_ * 3, 0.0);
      std::mt19937_64 gen(seed_);
      std::uniform_real_distribution<double> dis(-1.0, 1.0);
      for (size_t k = 0; k < (size_t)n_agents_ * n_obs_; ++k) {
        double x = dis(gen), y = dis(gen), z = dis(gen);
        // (makeRandomVector's normalized(), B/src/helper_functions.cpp:7-13: Eigen's squaredNorm in the association the
        // product library was built for -- compile the node with -DPMAF_DOT_RIGHT_ASSOC when it links the rassoc variant)
#ifdef PMAF_DOT_RIGHT_ASSOC
        double zz = x * x + (y * y + z * z);
#else
        double zz = (x * x + y * y) + z * z;
#endif
        if (zz > 0) { double s = std::sqrt(zz); x = x / s; y = y / s; z = z / s; }
        random_vecs_[k * 3] = x; random_vecs_[k * 3 + 1] = y; random_vecs_[k * 3 + 2] = z;
      }
      seed_ = gen();  // a later init() draws fresh vectors, like the reference
    }
    const std::vector<double> obs = flat(obstacles);
    const double goal[3] = {goal_pos.x(), goal_pos.y(), goal_pos.z()};
    const double ip[3] = {init_pos_.x(), init_pos_.y(), init_pos_.z()};
    pmaf_params prm{};
    prm.abi_version = PMAF_ABI_VERSION;
    prm.n_populations = 1;
    prm.n_agents = n_agents_;
    prm.n_obstacles = n_obs_;
    prm.max_prediction_steps = cap_;
    prm.device = device_;
    prm.lanes_per_agent = 0;
    prm.dt = (double)prediction_freq_multiple * delta_t;
    prm.velocity_max = velocity_max;
    prm.approach_dist = approach_dist;
    prm.detect_shell_rad = detect_shell_rad;
    prm.agent_mass = agent_mass;
    prm.radius = radius;
    prm.goal = goal;
    prm.init_pos = ip;
    prm.obstacles = obs.data();
    prm.k_attr = k_a_ee.data();
    prm.k_circ = k_c_ee.data();
    prm.k_repel = k_r_ee.data();
    prm.k_damp = k_d_ee.data();
    prm.agent_types = nullptr;  // Had, Goal, Obstacle, GoalObstacle, Vel, Random..., :70-104
    prm.random_vecs = random_vecs_.data();
    check(pmaf_create(&prm, &h_), "CfManager::init");
    if (best_id_ > 0 && best_id_ <= n_agents_ && best_rand_.size() == (size_t)n_obs_ * 3) {
      const int32_t id = best_id_, type = best_type_;
      check(pmaf_set_best(h_, &id, &type, best_rand_.data()), "CfManager::init(best)");
    }
    if (comm_) check(pmaf_attach_comm(h_, comm_), "CfManager::init(communicator)");
    if (selected_path_on_) check(pmaf_enable_winner_path(h_, 1), "CfManager::init(selected path)");
  }

  void startPrediction() { require(); touch(); check(pmaf_start(h_), "startPrediction"); }   // cf_manager.h:57-61
  void stopPrediction() { require(); check(pmaf_stop(h_), "stopPrediction"); }      // cf_manager.cpp:126-140
  void shutdownAllAgents() {}                                                       // no threads to stop
  void joinPredictionThreads() { if (h_) check(pmaf_stop(h_), "joinPredictionThreads"); }

  // cf_manager.cpp:184-190. Returned by const reference to the cached conversion (the reference returns a fresh
  // copy each time): `getPredictedPaths().size()`, `.at(i)`, `auto p = getPredictedPaths();` compile unchanged and
  // the node's 3 N + 1 calls per tick cost nothing. Valid until the next call that changes the predicted paths.
  const std::vector<std::vector<Vector3d>> &getPredictedPaths() {
    require();
    if (!paths_cached_) {
      const double *p = nullptr;
      const int32_t *n = nullptr;
      check(pmaf_view_paths(h_, &p, &n), "getPredictedPaths");   // the handle's host mirror: one D2H per rollout
      paths_cache_.resize(n_agents_);
      for (int a = 0; a < n_agents_; ++a) {
        std::vector<Vector3d> &out = paths_cache_[a];
        out.clear();
        out.reserve(n[a]);
        for (int k = 0; k < n[a]; ++k) {
          const double *q = &p[((size_t)a * cap_ + k) * 3];
          out.push_back(Vector3d(q[0], q[1], q[2]));
        }
      }
      paths_cached_ = true;
    }
    return paths_cache_;
  }
  std::vector<double> getPredictedPathLengths() {                                   // :192-198
    require();
    std::vector<double> out(n_agents_);
    check(pmaf_get_path_lengths(h_, out.data()), "getPredictedPathLengths");
    return out;
  }
  std::vector<double> getPredictionTimes() {                                        // :200-206
    require();
    std::vector<double> out(n_agents_, 0.0);
    if (pmaf_get_prediction_times_ns(h_, out.data()) != PMAF_OK) out.assign(n_agents_, 0.0);
    return out;
  }
  std::vector<bool> getAgentSuccess() {                                             // :208-214
    require();
    std::vector<int32_t> s(n_agents_);
    check(pmaf_get_success(h_, s.data()), "getAgentSuccess");
    return std::vector<bool>(s.begin(), s.end());
  }
  int getBestAgentType() {                                                          // cf_manager.h:73
    require();
    int32_t type = -1, id = 0;
    check(pmaf_get_best(h_, &type, &id), "getBestAgentType");
    if (id == 0) throw std::logic_error("getBestAgentType: no best agent yet");
    return type;
  }
  Vector3d getNextPosition() {                                                      // :74-76
    require();
    double p[3];
    check(pmaf_get_real_state(h_, p, nullptr, nullptr), "getNextPosition");
    return Vector3d(p[0], p[1], p[2]);
  }
  Vector3d getInitialPosition() { return init_pos_; }                               // :77
  Vector3d getNextVelocity() {                                                      // :78
    require();
    double v[3];
    check(pmaf_get_real_state(h_, nullptr, v, nullptr), "getNextVelocity");
    return Vector3d(v[0], v[1], v[2]);
  }
  Vector3d getEEForce() {                                                           // :79
    require();
    double f[3];
    check(pmaf_get_real_state(h_, nullptr, nullptr, f), "getEEForce");
    return Vector3d(f[0], f[1], f[2]);
  }
  Vector3d getGoalPosition() const { return goal_pos_; }                            // :80
  int getNumPredictionSteps(int agent_id) {                                         // :81-83
    require();
    if (agent_id < 0 || agent_id >= n_agents_) throw std::out_of_range("getNumPredictionSteps: agent index");
    const int32_t *n = nullptr;
    check(pmaf_view_paths(h_, nullptr, &n), "getNumPredictionSteps");
    return n[agent_id];
  }
  int getRealNumPredictionSteps() {                                                 // :84-86
    require();
    int32_t n = 0;
    check(pmaf_get_real_path(h_, 0, nullptr, 0, &n), "getRealNumPredictionSteps");
    return n;
  }
  double getDistFromGoal() const {                                                  // :87-89
    if (!h_) throw std::logic_error("CfManager: init() has not been called");
    double d = 0.0;
    check(pmaf_get_dist_from_goal(h_, &d), "getDistFromGoal");
    return d;
  }
  std::vector<Vector3d> getPlannedTrajectory() const {                              // :90-92
    if (!h_) throw std::logic_error("CfManager: init() has not been called");
    int32_t n = 0;
    check(pmaf_get_real_path(h_, 0, nullptr, 0, &n), "getPlannedTrajectory");
    std::vector<double> p((size_t)n * 3);
    check(pmaf_get_real_path(h_, 0, p.data(), n, nullptr), "getPlannedTrajectory");
    std::vector<Vector3d> out;
    out.reserve(n);
    for (int k = 0; k < n; ++k) out.push_back(Vector3d(p[k * 3], p[k * 3 + 1], p[k * 3 + 2]));
    return out;
  }
  // CfManager::getLinkForce, B/src/cf_manager.cpp:169-182
  std::vector<Vector3d> getLinkForce(const std::vector<Vector3d> &link_positions,
                                     const std::vector<Obstacle> &obstacles) {
    require();
    if (k_r_force_.size() != link_positions.size())
      throw std::invalid_argument("getLinkForce: one link position per k_r_force entry");  // assert, :172
    std::vector<double> lp(link_positions.size() * 3), out(link_positions.size() * 3);
    for (size_t i = 0; i < link_positions.size(); ++i) {
      lp[i * 3] = link_positions[i].x(); lp[i * 3 + 1] = link_positions[i].y(); lp[i * 3 + 2] = link_positions[i].z();
    }
    if ((int)obstacles.size() != n_obs_) throw std::out_of_range("getLinkForce: obstacle count changed");  // obstacles.back()
    const std::vector<double> obs = flat(obstacles);
    check(pmaf_link_force(h_, 0, (int32_t)link_positions.size(), lp.data(), k_r_force_.data(), obs.data(), out.data()),
          "getLinkForce");
    std::vector<Vector3d> forces;
    for (size_t i = 0; i < link_positions.size(); ++i)
      forces.push_back(Vector3d(out[i * 3], out[i * 3 + 1], out[i * 3 + 2]));
    return forces;
  }

  void setRealEEAgentPosition(const Vector3d &position) {                           // :216-218
    require();
    const double p[3] = {position.x(), position.y(), position.z()};
    check(pmaf_set_real_position(h_, p), "setRealEEAgentPosition");
  }
  void setInitialPosition(const Vector3d &position) {                               // :226-229
    init_pos_ = position;
    setInitialEEPositions(position);
  }
  void setInitialEEPositions(const Vector3d &position) {                            // :231-236
    if (!h_) return;  // default-constructed manager: no agents yet (empty loops in the reference)
    touch();
    const double p[3] = {position.x(), position.y(), position.z()};
    check(pmaf_set_initial_position(h_, p), "setInitialPosition");
  }
  // CfManager::resetEEAgents, :246-255
  void resetEEAgents(const Vector3d &position, const Vector3d &velocity, const std::vector<Obstacle> &obstacles) {
    require();
    const double p[3] = {position.x(), position.y(), position.z()};
    const double v[3] = {velocity.x(), velocity.y(), velocity.z()};
    const std::vector<double> obs = flat(obstacles);
    if ((int)obstacles.size() != n_obs_) throw std::out_of_range("resetEEAgents: obstacle count changed");  // .at(), cf_agent.cpp:66
    touch();
    check(pmaf_reset_agents(h_, p, v, obs.data()), "resetEEAgents");
  }
  // CfManager::moveRealEEAgent, :257-263
  void moveRealEEAgent(const std::vector<Obstacle> &obstacles, const double delta_t, const int steps,
                       const int agent_id) {
    require();
    const std::vector<double> obs = flat(obstacles);
    if ((int)obstacles.size() != n_obs_) throw std::out_of_range("moveRealEEAgent: obstacle count changed");
    const int32_t id = agent_id;
    check(pmaf_move_real(h_, obs.data(), delta_t, steps, &id), "moveRealEEAgent");
  }
  // CfManager::evaluateAgents, :293-356 (its obstacles argument is unused there too)
  int evaluateAgents(const std::vector<Obstacle> &obstacles, const double k_goal_dist, const double k_path_len,
                     const double k_safe_dist, const double k_workspace, const Vector6d des_ws_limits) {
    (void)obstacles;
    require();
    const double gains[4] = {k_goal_dist, k_path_len, k_safe_dist, k_workspace};
    double ws[6];
    for (int i = 0; i < 6; ++i) ws[i] = des_ws_limits(i);
    int32_t best = 0;
    check(pmaf_evaluate(h_, gains, ws, &best), "evaluateAgents");
    return best;
  }
  // the whole planCallback sequence (B/src/panda_bimanual_control.cpp:336-352)
  // as one call: stop, evaluate, move the real agent one step, reset, start
  int planTick(const std::vector<Obstacle> &obstacles, const double delta_t, const double k_goal_dist,
               const double k_path_len, const double k_safe_dist, const double k_workspace,
               const Vector6d des_ws_limits, Vector3d *next_position = nullptr) {
    require();
    if ((int)obstacles.size() != n_obs_) throw std::out_of_range("planTick: obstacle count changed");
    touch();
    const std::vector<double> obs = flat(obstacles);
    const double gains[4] = {k_goal_dist, k_path_len, k_safe_dist, k_workspace};
    double ws[6];
    for (int i = 0; i < 6; ++i) ws[i] = des_ws_limits(i);
    int32_t best = 0;
    double np[3];
    check(pmaf_tick(h_, obs.data(), delta_t, gains, ws, &best, np, nullptr), "planTick");
    if (next_position) *next_position = Vector3d(np[0], np[1], np[2]);
    // failure detection (pmaf.h, PMAF_HEALTH_*), opt-in (setThrowOnNumericFault): a NaN / infinite set-point reported as
    // an exception instead of being published like the reference does. getHealth() reports it either way, also for the
    // individual calls of the reference's surface (moveRealEEAgent ...).
    if (throw_on_numeric_fault_ && (getHealth() & (PMAF_HEALTH_SETPOINT_NAN | PMAF_HEALTH_FORCE_NAN)))
      throw std::runtime_error("CfManager::planTick: the real agent's set-point / force is not finite (getHealth())");
    return best;
  }
  // PMAF_HEALTH_* bits of the last planTick / evaluateAgents / moveRealEEAgent (no reference equivalent)
  int getHealth() {
    require();
    int32_t b = 0;
    check(pmaf_get_health(h_, &b), "getHealth");
    return (int)b;
  }
  void setThrowOnNumericFault(bool on) { throw_on_numeric_fault_ = on; }
  // The selected trajectory of every planTick / evaluateAgents on the host without the copy of all N paths
  // getPredictedPaths() makes (what the node's visualisation marks as the best path,
  // B/src/panda_bimanual_control.cpp:340-347): enable once, then getSelectedPath() after a tick returns the path the
  // selection scored (pmaf_view_winner_path: written by the manager kernel into pinned memory behind the set-point).
  void enableSelectedPath(bool on = true) {
    require();
    selected_path_on_ = on;
    check(pmaf_enable_winner_path(h_, on ? 1 : 0), "enableSelectedPath");
  }
  std::vector<Vector3d> getSelectedPath(int *agent_index = nullptr) {
    require();
    const double *p = nullptr;
    const int32_t *n = nullptr, *a = nullptr;
    check(pmaf_view_winner_path(h_, &p, &n, &a), "getSelectedPath");
    std::vector<Vector3d> out;
    out.reserve((size_t)n[0]);
    for (int k = 0; k < n[0]; ++k) out.emplace_back(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
    if (agent_index) *agent_index = a[0];
    return out;
  }

  // ---- synchronous stepping API (no callers in the reference; B/src/cf_manager.cpp:220-224, 238-244, 265-291) ----
  // Deviation: after any of these calls startPrediction() needs a resetEEAgents() / setInitialPosition() first
  // (std::runtime_error otherwise): rollouts start from the population's reset state, the reference's threads would
  // continue with each agent's own velocity, known flags and advanced obstacle copies (pmaf.h).
  void setEEAgentPositions(const Vector3d &position) {                              // :220-224
    require();
    touch();
    const double p[3] = {position.x(), position.y(), position.z()};
    check(pmaf_set_agent_positions(h_, p), "setEEAgentPositions");
  }
  void setEEAgentPosAndVels(const Vector3d &position, const Vector3d &velocity) {   // :238-244
    require();
    touch();
    const double p[3] = {position.x(), position.y(), position.z()};
    const double v[3] = {velocity.x(), velocity.y(), velocity.z()};
    check(pmaf_set_agent_pos_and_vels(h_, p, v), "setEEAgentPosAndVels");
  }
  // moveAgent (:265-272): cfPlanner(steps) while the agent is farther than 0.05 from the goal -- bounded here by
  // the path buffer (max_prediction_steps points)
  void moveAgent(const std::vector<Obstacle> &obstacles, const double delta_t, const int steps, const int id) {
    require();
    if ((int)obstacles.size() != n_obs_) throw std::out_of_range("moveAgent: obstacle count changed");
    if (id < 0 || id >= n_agents_) throw std::out_of_range("moveAgent: agent index");
    touch();
    const std::vector<double> obs = flat(obstacles);
    const int32_t a = id;
    check(pmaf_move_agent(h_, obs.data(), delta_t, steps, &a, 0x7fffffff, nullptr), "moveAgent");
  }
  void moveAgents(const std::vector<Obstacle> &obstacles, const double delta_t, const int steps = 1) {  // :274-282
    require();
    if ((int)obstacles.size() != n_obs_) throw std::out_of_range("moveAgents: obstacle count changed");
    touch();
    const std::vector<double> obs = flat(obstacles);
    check(pmaf_move_agents(h_, obs.data(), delta_t, steps), "moveAgents");
  }
  void moveAgentsPar(const std::vector<Obstacle> &obstacles, const double delta_t, const int steps = 1) {  // :284-291
    moveAgents(obstacles, delta_t, steps);  // every agent is its own wave / lane group on the device anyway
  }
  // CfAgent::evalObstacleDistance (B/src/cf_agent.cpp:146-157) of every predicted agent (the reference exposes it
  // per agent object; the agents live on the device here)
  std::vector<double> evalObstacleDistances(const std::vector<Obstacle> &obstacles) {
    require();
    if ((int)obstacles.size() != n_obs_) throw std::out_of_range("evalObstacleDistances: obstacle count changed");
    const std::vector<double> obs = flat(obstacles);
    std::vector<double> out(n_agents_);
    check(pmaf_eval_obstacle_distance(h_, obs.data(), out.data()), "evalObstacleDistances");
    return out;
  }

 private:
  std::vector<double> random_vecs_override_;
};

}  // namespace cfplanner
}  // namespace ghostplanner

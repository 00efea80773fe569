// planner_node.h -- ROS-free mirror of the layers directly above and beside
// the planner tick, in C++ like the reference:
//
//   TaskParams / loadTaskFile   the `bimanual_planning:` keys of the task YAML
//                               files (B/config/tasks/*.yaml, registered at
//                               B/src/panda_bimanual_control.cpp:129-170) incl.
//                               per-goal overrides of planner keys (:193);
//   Position / Obstacles        the wire layouts of B/msg/Position.msg and
//                               B/msg/Obstacles.msg;
//   PlannerNode                 the planner half of PandaBimanualPlanning:
//                               planCallback (:329-369), obstacleCallback
//                               (:302-309), the PLAN branch of taskCallback
//                               (:494-522) and the REACHED end condition
//                               (:565-569), driving CfManager (the facade over
//                               libpmaf_hip.so);
//   DynamicObstacleSource       the integration loop of dynamic_obstacle_node
//                               (B/src/dynamic_obstacle_node.cpp:352-383):
//                               cur_pos += cur_vel / frequency for the first M
//                               obstacles, published as an Obstacles message;
//   SetPointConsumer            (setpoint_consumer.h) the receiving side of the
//                               "goals" topic: TrajectoryBuffer + the trajectory
//                               half of CoSTPController::followTrajectory
//                               (B/src/costp_controller.cpp:289-344,
//                               B/src/trajectory_buffer.cpp:13-64) and the
//                               hand-over loop of VrepController
//                               (B/src/vrep_controller.cpp:100-115);
//   validateSetPoint            the stateless part of that contract (finite,
//                               >= 1e-6 m from the previous point).
// B/ = reference src/bimanual_planning_ros/. No ROS types: topics become plain
// function calls so the loop can run head-less (tools/plan_task.cpp).
#pragma once

#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "bimanual_planning_ros/cf_manager.h"
#include "bimanual_planning_ros/setpoint_consumer.h"

namespace ghostplanner {
namespace cfplanner {

// ---- message layouts ------------------------------------------------------
struct Position { double data[3]; };                    // B/msg/Position.msg:1  float64[3] data
struct Obstacles {                                      // B/msg/Obstacles.msg:1-3
  std::vector<Position> pos, vel;
  std::vector<double> radius;
};

// ---- minimal YAML subset (block maps / sequences, flow lists, scalars) ----
namespace yaml_lite {
struct Node {
  enum Kind { NONE, SCALAR, MAP, SEQ } kind = NONE;
  std::string scalar;
  std::vector<std::pair<std::string, Node>> map;
  std::vector<Node> seq;
  const Node *find(const std::string &k) const {
    for (auto &kv : map) if (kv.first == k) return &kv.second;
    return nullptr;
  }
  double num() const { return std::stod(scalar); }
  bool boolean() const { return scalar == "true" || scalar == "True" || scalar == "1"; }
  std::vector<double> nums() const {
    std::vector<double> v;
    for (auto &n : seq) v.push_back(n.num());
    return v;
  }
};
struct Line { int indent; std::string text; };
inline std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
  return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
inline std::string unquote(std::string s) {
  s = trim(s);
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) s = s.substr(1, s.size() - 2);
  return s;
}
inline Node parse_value(const std::string &txt) {
  Node n;
  std::string t = trim(txt);
  if (!t.empty() && t.front() == '[') {
    n.kind = Node::SEQ;
    std::string inner = t.substr(1, t.rfind(']') - 1), item;
    std::stringstream ss(inner);
    while (std::getline(ss, item, ',')) {
      if (trim(item).empty()) continue;
      Node c; c.kind = Node::SCALAR; c.scalar = unquote(item);
      n.seq.push_back(c);
    }
  } else {
    n.kind = Node::SCALAR;
    n.scalar = unquote(t);
  }
  return n;
}
inline Node parse_block(const std::vector<Line> &L, size_t &i, int indent);
inline void parse_map_entry(const std::vector<Line> &L, size_t &i, int indent, const std::string &text, Node &m) {
  size_t c = text.find(':');
  if (c == std::string::npos) throw std::runtime_error("yaml: expected key: value in '" + text + "'");
  std::string key = trim(text.substr(0, c)), rest = trim(text.substr(c + 1));
  ++i;
  if (!rest.empty()) { m.map.emplace_back(key, parse_value(rest)); return; }
  if (i < L.size() && L[i].indent > indent) m.map.emplace_back(key, parse_block(L, i, L[i].indent));
  else if (i < L.size() && L[i].indent == indent && L[i].text.rfind("- ", 0) == 0) m.map.emplace_back(key, parse_block(L, i, indent));
  else m.map.emplace_back(key, Node());
}
inline Node parse_block(const std::vector<Line> &L, size_t &i, int indent) {
  Node n;
  if (i >= L.size()) return n;
  if (L[i].text.rfind("- ", 0) == 0 || L[i].text == "-") {
    n.kind = Node::SEQ;
    while (i < L.size() && L[i].indent == indent && (L[i].text.rfind("- ", 0) == 0 || L[i].text == "-")) {
      std::string rest = trim(L[i].text.substr(1));
      if (rest.find(':') != std::string::npos && rest.front() != '[') {
        // "- key: value" starts a map whose further keys are indented by two more columns
        Node item; item.kind = Node::MAP;
        int child = indent + 2;
        std::vector<Line> first{{child, rest}};
        size_t j = 0;
        parse_map_entry(first, j, child, rest, item);
        ++i;
        while (i < L.size() && L[i].indent >= child && !(L[i].indent == indent)) {
          if (L[i].indent != child) throw std::runtime_error("yaml: bad indentation near '" + L[i].text + "'");
          parse_map_entry(L, i, child, L[i].text, item);
        }
        n.seq.push_back(item);
      } else {
        n.seq.push_back(parse_value(rest));
        ++i;
      }
    }
  } else {
    n.kind = Node::MAP;
    while (i < L.size() && L[i].indent == indent && L[i].text.rfind("- ", 0) != 0) parse_map_entry(L, i, indent, L[i].text, n);
  }
  return n;
}
inline Node parse(std::istream &in) {
  std::vector<Line> L;
  std::string raw;
  while (std::getline(in, raw)) {
    size_t h = raw.find('#');
    if (h != std::string::npos) raw = raw.substr(0, h);
    if (trim(raw).empty()) continue;
    int ind = 0;
    while (ind < (int)raw.size() && raw[ind] == ' ') ++ind;
    L.push_back({ind, trim(raw)});
  }
  size_t i = 0;
  return L.empty() ? Node() : parse_block(L, i, L[0].indent);
}
}  // namespace yaml_lite

// ---- task parameters --------------------------------
struct GoalSpec {
  std::string type, end_condition, message;
  Vector3d pos{0, 0, 0};
  std::map<std::string, double> overrides;  // planner keys overridden for this goal (goal_pm_, optional = true)
};
struct TaskParams {
  int num_agents_ee = 10, num_agents_body = 1;
  double k_attr = 4.0, k_circ = 0.025, k_repel = 0.08, k_damp = 3.0, k_manip = 0.0, k_repel_body = 0.02;
  double k_goal_dist = 100.0, k_path_len = 10.0, k_safe_dist = 0.001, k_workspace = 1.0;
  Vector6d desired_ws_limits;
  int max_prediction_steps = 1500, prediction_freq_multiple = 1;
  double approach_dist = 0.25, detect_shell_rad = 0.35, frequency_ros = 100.0, velocity = 0.2;
  bool open_loop = true, visualize_commanded_path = true, visualize_predicted_paths = true;
  std::vector<Obstacle> obstacles;
  std::vector<GoalSpec> goals;
  bool setScalar(const std::string &k, double v) {
    if (k == "num_agents_ee") num_agents_ee = (int)v; else if (k == "num_agents_body") num_agents_body = (int)v;
    else if (k == "k_attr") k_attr = v; else if (k == "k_circ") k_circ = v; else if (k == "k_repel") k_repel = v;
    else if (k == "k_damp") k_damp = v; else if (k == "k_manip") k_manip = v; else if (k == "k_repel_body") k_repel_body = v;
    else if (k == "k_goal_dist") k_goal_dist = v; else if (k == "k_path_len") k_path_len = v;
    else if (k == "k_safe_dist") k_safe_dist = v; else if (k == "k_workspace") k_workspace = v;
    else if (k == "max_prediction_steps") max_prediction_steps = (int)v;
    else if (k == "prediction_freq_multiple") prediction_freq_multiple = (int)v;
    else if (k == "approach_dist") approach_dist = v; else if (k == "detect_shell_rad") detect_shell_rad = v;
    else if (k == "frequency_ros") frequency_ros = v; else if (k == "velocity") velocity = v;
    else return false;
    return true;
  }
};

inline TaskParams parseTask(std::istream &in) {
  using yaml_lite::Node;
  Node root = yaml_lite::parse(in);
  const Node *bp = root.find("bimanual_planning");  // B/src/panda_bimanual_control.cpp:433-441
  if (!bp || bp->kind != Node::MAP) throw std::runtime_error("task file: missing 'bimanual_planning' map");
  TaskParams t;
  for (int i = 0; i < 6; ++i) t.desired_ws_limits(i) = (i % 2 == 0) ? 1e300 : -1e300;
  for (auto &kv : bp->map) {
    const Node &v = kv.second;
    if (v.kind == Node::SCALAR) {
      if (kv.first == "open_loop") t.open_loop = v.boolean();
      else if (kv.first == "visualize_commanded_path") t.visualize_commanded_path = v.boolean();
      else if (kv.first == "visualize_predicted_paths") t.visualize_predicted_paths = v.boolean();
      else { try { t.setScalar(kv.first, v.num()); } catch (const std::exception &) {} }
    } else if (kv.first == "desired_ws_limits") {
      std::vector<double> w = v.nums();
      if (w.size() != 6) throw std::runtime_error("task file: desired_ws_limits needs 6 entries");
      for (int i = 0; i < 6; ++i) t.desired_ws_limits(i) = w[i];
    } else if (kv.first == "obstacles") {          // :49-60
      for (auto &o : v.seq) {
        const Node *p = o.find("pos"), *r = o.find("radius"), *ve = o.find("vel");
        if (!p || !r) throw std::runtime_error("task file: obstacle needs pos and radius");
        std::vector<double> pp = p->nums(), vv = ve ? ve->nums() : std::vector<double>{0, 0, 0};
        t.obstacles.push_back(Obstacle(Vector3d(pp.at(0), pp.at(1), pp.at(2)), Vector3d(vv.at(0), vv.at(1), vv.at(2)), r->num()));
      }
    } else if (kv.first == "goals") {              // :194-217
      for (auto &g : v.seq) {
        GoalSpec gs;
        for (auto &gk : g.map) {
          if (gk.first == "type") gs.type = gk.second.scalar;
          else if (gk.first == "end_condition") gs.end_condition = gk.second.scalar;
          else if (gk.first == "message") gs.message = gk.second.scalar;
          else if (gk.first == "pos") { std::vector<double> pp = gk.second.nums(); gs.pos = Vector3d(pp.at(0), pp.at(1), pp.at(2)); }
          else if (gk.second.kind == Node::SCALAR) { try { gs.overrides[gk.first] = gk.second.num(); } catch (const std::exception &) {} }
        }
        t.goals.push_back(gs);
      }
    }
  }
  if (t.obstacles.empty()) throw std::runtime_error("task file: obstacle list must end with the repulsive self-collision obstacle");
  return t;
}
inline TaskParams loadTaskFile(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open task file " + path);
  return parseTask(f);
}

// ---- set-point consumer contract (f4) --------------------------------------
// CoSTPController::fillBuffer / followTrajectory accept a set-point only if it
// is finite (B/src/costp_controller.cpp:317-319) and at least 1e-6 m away from
// the previous one (followTrajectory nudges closer points, B/src/costp_controller.cpp:320-324).
inline bool validateSetPoint(const Vector3d &prev, const Vector3d &next, std::string *why = nullptr) {
  for (int i = 0; i < 3; ++i)
    if (!std::isfinite(next[i])) { if (why) *why = "non-finite set-point"; return false; }
  const double dx = next[0] - prev[0], dy = next[1] - prev[1], dz = next[2] - prev[2];
  if (std::sqrt(dx * dx + dy * dy + dz * dz) < 1e-6) { if (why) *why = "set-point closer than 1e-6 m to the previous one"; return false; }
  return true;
}

// ---- obstacle stream (f2) ---------------------------------------------------
class DynamicObstacleSource {
  std::vector<Vector3d> cur_pos_, cur_vel_;
  std::vector<double> radius_;
  double frequency_;

 public:
  DynamicObstacleSource(const std::vector<Obstacle> &obstacles, double frequency = 100.0) : frequency_(frequency) {
    for (size_t i = 0; i + 1 < obstacles.size(); ++i) {  // the trailing repulsive obstacle is not streamed, :317
      cur_pos_.push_back(obstacles[i].getPosition());
      cur_vel_.push_back(obstacles[i].getVelocity());
      radius_.push_back(obstacles[i].getRadius());
    }
  }
  Obstacles message() const {
    Obstacles m;
    for (size_t i = 0; i < cur_pos_.size(); ++i) {
      m.pos.push_back({{cur_pos_[i][0], cur_pos_[i][1], cur_pos_[i][2]}});
      m.vel.push_back({{cur_vel_[i][0], cur_vel_[i][1], cur_vel_[i][2]}});
      m.radius.push_back(radius_[i]);
    }
    return m;
  }
  // one iteration of the node's 100 Hz loop: cur_pos += cur_vel / frequency, :355-357
  Obstacles step() {
    for (size_t i = 0; i < cur_pos_.size(); ++i)
      for (int c = 0; c < 3; ++c) cur_pos_[i][c] = cur_pos_[i][c] + cur_vel_[i][c] / frequency_;
    return message();
  }
};

// ---- planner node (f1) --------------------------------------------------------
class PlannerNode {
  TaskParams prm_;
  CfManager cf_manager_;
  std::vector<Obstacle> obstacles_;
  Vector3d last_goal_{0, 0, 0};
  double time_step_ = 0.01;
  bool planning_active_ = false, got_initial_pos_ = false;
  std::vector<Vector3d> commanded_path_;

  void initManager() {                                 // :463-471 / :501-509
    const int n = prm_.num_agents_ee;
    cf_manager_.init(last_goal_, time_step_, obstacles_, std::vector<double>(n, prm_.k_attr),
                     std::vector<double>(n, prm_.k_circ), std::vector<double>(n, prm_.k_repel),
                     std::vector<double>(n, prm_.k_damp), std::vector<double>(n, prm_.k_manip),
                     std::vector<double>(prm_.num_agents_body, prm_.k_repel_body), prm_.velocity, prm_.approach_dist,
                     prm_.detect_shell_rad, (size_t)prm_.max_prediction_steps, (size_t)prm_.prediction_freq_multiple);
  }

 public:
  explicit PlannerNode(const TaskParams &p, uint64_t random_seed = 1, int device = -1) : prm_(p), obstacles_(p.obstacles) {
    time_step_ = 1.0 / prm_.frequency_ros;             // :78-80
    cf_manager_.setRandomSeed(random_seed);
    cf_manager_.setDevice(device);
    initManager();                                     // node start-up, :463-471
  }
  CfManager &manager() { return cf_manager_; }
  const std::vector<Obstacle> &obstacles() const { return obstacles_; }
  bool planningActive() const { return planning_active_; }
  double timeStep() const { return time_step_; }

  // obstacleCallback, :302-309
  void obstacleCallback(const Obstacles &msg) {
    for (size_t i = 0; i < msg.radius.size(); ++i) {
      obstacles_.at(i).setPosition(Vector3d(msg.pos[i].data[0], msg.pos[i].data[1], msg.pos[i].data[2]));
      obstacles_.at(i).setVelocity(Vector3d(msg.vel[i].data[0], msg.vel[i].data[1], msg.vel[i].data[2]));
    }
  }

  // taskCallback, GoalType::PLAN branch (:494-522): re-init towards the goal,
  // start from the current position; returns the first published set-point
  Position startPlan(const GoalSpec &goal) {
    if (!got_initial_pos_) throw std::logic_error("startPlan: no initial position yet (planCallback must run first)");
    for (auto &kv : goal.overrides) prm_.setScalar(kv.first, kv.second);
    time_step_ = 1.0 / prm_.frequency_ros;
    last_goal_ = goal.pos;
    Vector3d current_pos = cf_manager_.getNextPosition();
    initManager();
    cf_manager_.setInitialPosition(current_pos);
    commanded_path_.clear();
    planning_active_ = true;
    Vector3d ip = cf_manager_.getInitialPosition();
    return Position{{ip[0], ip[1], ip[2] + 0.00001}};  // :514-518
  }

  // planCallback, :329-369. Returns true and fills `out` with the next
  // set-point ("goals" topic) while planning is active; otherwise records the
  // initial position (:364-367) and returns false.
  bool planCallback(const Position &p, Position *out, int *best_agent = nullptr) {
    if (!planning_active_) {
      cf_manager_.setInitialPosition(Vector3d(p.data[0], p.data[1], p.data[2]));
      got_initial_pos_ = true;
      return false;
    }
    if (!prm_.open_loop) cf_manager_.setRealEEAgentPosition(Vector3d(p.data[0], p.data[1], p.data[2]));
    Vector3d next;
    const int best = cf_manager_.planTick(obstacles_, time_step_, prm_.k_goal_dist, prm_.k_path_len, prm_.k_safe_dist,
                                          prm_.k_workspace, prm_.desired_ws_limits, &next);
    if (best_agent) *best_agent = best;
    if (out) *out = Position{{next[0], next[1], next[2]}};
    if (prm_.visualize_commanded_path) commanded_path_.push_back(next);
    return true;
  }
  // EndCondition::REACHED, :565-569
  bool reached() const { return cf_manager_.getDistFromGoal() < 0.01; }
  void finishGoal() { planning_active_ = false; }     // :583-587
  double goalDistance() const { return cf_manager_.getDistFromGoal(); }  // "goal_distance" topic, :358-360
  const std::vector<std::vector<Vector3d>> &predictedPaths() { return cf_manager_.getPredictedPaths(); }  // :340-347
  double velocity() const { return prm_.velocity; }
  const std::vector<Vector3d> &commandedPath() const { return commanded_path_; }                   // :361-363
};

}  // namespace cfplanner
}  // namespace ghostplanner

// setpoint_consumer.h -- what the planner's product (the "goals" topic) runs into on the controller side
// (SURVEY.md 8f row f4), restated ROS- and robot-free so that the head-less loop (tools/plan_task.cpp) can check
// every set-point it emits against the consumer's real acceptance logic:
//
//   SetPointHandOver    the one-slot contract of the controller's set-point buffer (the reference's TrajectoryBuffer,
//                       B/src/trajectory_buffer.cpp, is used with size 1: B/src/costp_controller.cpp:25): a second
//                       set-point before the first was taken is refused.
//   SetPointConsumer    the 3-vector half of CoSTPController (B/ = reference src/bimanual_planning_ros/):
//                         reset            B/src/costp_controller.cpp:88-109
//                         fillBuffer       :289-297
//                         followTrajectory :299-344 (acceptance: NaN check, the < 1e-6 m nudge of +-2e-6 in z,
//                                          v_goal = min(|d| * 100, (1 - reserve) * v_max), the radicand >= 0
//                                          "inconsistent trajectory" test, next_ng)
//                         absolutePositionControl's speed ramp :193-201, getInstantaneousGoal :144-155,
//                         getCurrentNominalGoal :138-142
//                       i.e. everything between a received set-point and the instantaneous goal handed to the
//                       joint-space position controller. The joint-space control itself (dqrobotics kinematics,
//                       Jacobians) is out of scope; update() returns the instantaneous goal instead of joint angles.
//   handshake           VrepController::targetPoseCallback (B/src/vrep_controller.cpp:85-115): fillBuffer, then the
//                       1 kHz update loop until readyForNextPoint(); v_max handed to followTrajectory is
//                       velocity / 0.9 (B/src/vrep_controller.cpp:291-292).
//
// Arithmetic follows the reference's Eigen expressions in the oracle's conventions (norm = sqrt((x*x + y*y) + z*z),
// normalized() = component-wise division unless the squared norm is 0, std::pow(x, 2) = x * x), so the C++ class and
// the oracle's restatement (oracle/pmaf_oracle.c: orc_consumer_*) agree bit for bit.
#pragma once

#include <cmath>
#include <vector>

#include "bimanual_planning_ros/obstacle.h"

namespace ghostplanner {
namespace cfplanner {

// The hand-over between the "goals" subscriber and the controller cycle. The reference uses a ring buffer class
// (B/src/trajectory_buffer.cpp) with size 1 (B/src/costp_controller.cpp:25); what the planner's output has to live
// with is that size-1 CONTRACT, stated here instead of the class:
//   * ONE slot: a set-point offered while the previous one has not been taken is REFUSED (the reference logs
//     "Couldn't put trajectory point into buffer" and drops it);
//   * taking the set-point empties the slot; the controller asks for the next one (readyForNextPoint) only then.
struct SetPointHandOver {
  bool occupied = false;
  Vector3d point;
  bool offer(const Vector3d &p) {
    if (occupied) return false;
    point = p;
    occupied = true;
    return true;
  }
  bool take(Vector3d &p) {
    if (!occupied) return false;
    p = point;
    occupied = false;
    return true;
  }
  void clear() { occupied = false; }
};

class SetPointConsumer {
 public:
  struct Counters {
    long accepted = 0;        // points taken out of the buffer
    long refused = 0;         // fillBuffer on a full buffer ("Couldn't put trajectory point into buffer")
    long nan = 0;             // "Planner sent NaN."
    long too_close = 0;       // "Points sent by the planner are too close together." (nudged)
    long inconsistent = 0;    // "Inconsistent trajectory detected."
    long updates = 0;         // 1 kHz controller cycles
  };

 private:
  struct V { double x, y, z; };
  static V mk(const Vector3d &a) { return V{a[0], a[1], a[2]}; }
  static Vector3d out(V a) { return Vector3d(a.x, a.y, a.z); }
  static V add(V a, V b) { return V{a.x + b.x, a.y + b.y, a.z + b.z}; }
  static V sub(V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
  static V mul(double s, V a) { return V{s * a.x, s * a.y, s * a.z}; }
  static V mulr(V a, double s) { return V{a.x * s, a.y * s, a.z * s}; }
#ifdef PMAF_DOT_RIGHT_ASSOC   // the evaluation-order policy of the library this is built against (include/pmaf.h, pmaf_eval_order)
  static double dot(V a, V b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
#else
  static double dot(V a, V b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
#endif
  static double norm(V a) { return std::sqrt(dot(a, a)); }
  static V normalized(V a) {
    const double z = dot(a, a);
    if (z > 0.0) { const double s = std::sqrt(z); return V{a.x / s, a.y / s, a.z / s}; }
    return a;
  }

  SetPointHandOver slot_;                        // the size-1 buffer of costp_controller.cpp:25, as a contract
  bool ready_for_next_point_ = false;            // costp_controller.h:82
  V lg_{0, 0, 0}, cg_{0, 0, 0}, current_ng_{0, 0, 0}, last_ng_{0, 0, 0}, next_ig_{0, 0, 0}, current_ig_{0, 0, 0};
  double next_ng_ = 0.0, v_act_ = 0.0, v_goal_ = 0.0;
  double catchup_reserve_ = 0.1;                 // costp_controller.h:79
  double min_motion_ = 2e-6;                     // followTrajectory's function-local static, :301
  Counters cnt_;

  void getCurrentNominalGoal() {                 // :138-142
    last_ng_ = current_ng_;
    current_ng_ = add(lg_, mul(next_ng_, sub(cg_, lg_)));
    next_ng_ += v_goal_ * 0.001 / norm(sub(cg_, lg_));
  }
  V getInstantaneousGoal() {                     // :144-155
    const V current_ig = next_ig_;
    if (norm(sub(current_ng_, current_ig)) < v_act_ * 0.001) getCurrentNominalGoal();
    next_ig_ = add(current_ig, mul(v_act_ * 0.001, normalized(sub(current_ng_, current_ig))));
    current_ig_ = add(mul(0.9, current_ig_), mul(0.1, current_ig));
    return current_ig_;
  }

 public:
  SetPointConsumer() = default;
  // CoSTPController::reset, :88-109, with the end-effector position the forward kinematics would return
  void reset(const Vector3d &ee_position) {
    ready_for_next_point_ = true;
    slot_.clear();
    lg_ = mk(ee_position);
    next_ig_ = lg_; cg_ = lg_; current_ng_ = lg_; last_ng_ = lg_;
    next_ng_ = 0;
    current_ig_ = lg_;
    v_act_ = 0;
    v_goal_ = 0;
  }
  bool readyForNextPoint() const { return ready_for_next_point_; }     // costp_controller.h:68
  // CoSTPController::fillBuffer, :289-297. false = the buffer refused the point (logged as an error there).
  bool fillBuffer(const Vector3d &goal) {
    bool ok = slot_.offer(goal);
    if (!ok) cnt_.refused++;
    if (slot_.occupied) ready_for_next_point_ = false;
    return ok;
  }
  // One controller cycle (1 kHz): followTrajectory's trajectory logic (:299-340), then the speed ramp of
  // absolutePositionControl (:193-201) and getInstantaneousGoal (:144-155). Returns the instantaneous goal.
  Vector3d update(double v_max) {
    cnt_.updates++;
    if (next_ng_ >= 1 || (v_act_ == 0 && next_ng_ == 0)) {
      bool got_point = false;
      lg_ = cg_;
      Vector3d taken;
      if (slot_.take(taken)) {
        cg_ = mk(taken);
        got_point = true;
      } else {
        next_ng_ = 0;
        v_act_ = 0;
      }
      ready_for_next_point_ = true;
      if (got_point) {
        cnt_.accepted++;
        if (std::isnan(cg_.x) || std::isnan(cg_.y) || std::isnan(cg_.z)) cnt_.nan++;
        if (norm(sub(cg_, lg_)) < 1e-6) {
          cnt_.too_close++;
          cg_.z += min_motion_;
          min_motion_ = -min_motion_;
        }
        const double v = v_max * (1 - catchup_reserve_);
        const double dist100 = norm(sub(cg_, lg_)) * 100;
        v_goal_ = (v < dist100) ? v : dist100;   // std::min(|d| * 100, v)
        const double acos_gamma = dot(normalized(sub(cg_, lg_)), sub(current_ng_, lg_));
        const double e = v_goal_ * 0.001;
        const double l = norm(sub(lg_, current_ng_));
        const double radicand = (acos_gamma * acos_gamma + e * e) - l * l;
        double b;
        if (radicand >= 0) {
          b = acos_gamma + std::sqrt(radicand);
        } else {
          b = 0;
          cnt_.inconsistent++;
        }
        next_ng_ = b / norm(sub(cg_, lg_));
      }
    }
    // control() -> absolutePositionControl, :193-201
    if (v_goal_ > v_act_) {
      v_act_ += 0.001 * 0.05;
      if (v_act_ > v_goal_) v_act_ = v_goal_;
    } else {
      v_act_ = v_goal_;
    }
    return out(getInstantaneousGoal());
  }
  // VrepController::targetPoseCallback, B/src/vrep_controller.cpp:100-115: hand one set-point over and cycle the
  // controller until it asks for the next one. Returns the number of 1 kHz cycles spent (bounded by max_cycles).
  long deliver(const Vector3d &set_point, double velocity, long max_cycles = 1000000) {
    fillBuffer(set_point);
    long n = 0;
    while (n < max_cycles) {
      update(velocity / 0.9);                    // B/src/vrep_controller.cpp:291-292
      ++n;
      if (ready_for_next_point_) break;
    }
    return n;
  }
  const Counters &counters() const { return cnt_; }
  double vGoal() const { return v_goal_; }
  double vAct() const { return v_act_; }
  double nextNg() const { return next_ng_; }
  Vector3d currentGoal() const { return out(cg_); }
  Vector3d lastGoal() const { return out(lg_); }
  Vector3d nominalGoal() const { return out(current_ng_); }
  Vector3d instantaneousGoal() const { return out(current_ig_); }
};

}  // namespace cfplanner
}  // namespace ghostplanner

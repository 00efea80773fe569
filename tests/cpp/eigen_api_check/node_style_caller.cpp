// API-shape check (see eigen3/Eigen/Dense next to this file): the call FORMS the reference's planner node uses on
// CfManager with Eigen types -- written here from the survey of B/src/panda_bimanual_control.cpp:329-369 (planCallback)
// and :463-471 (init) -- must compile against the facade's PMAF_USE_EIGEN branch. Syntax check only, never linked.
#include <array>
#include <vector>

#include "bimanual_planning_ros/cf_manager.h"

using namespace ghostplanner::cfplanner;
using Eigen::Vector3d;

struct PositionMsg { std::array<double, 3> data; };   // bimanual_planning_ros/Position: float64[3]

double node_style_tick(CfManager &cf_manager_, const PositionMsg &p, std::vector<Obstacle> &obstacles_, bool open_loop_,
                       double time_step_, PositionMsg &out) {
  if (!open_loop_) cf_manager_.setRealEEAgentPosition(Vector3d(p.data.data()));          // :334
  cf_manager_.stopPrediction();
  Eigen::Matrix<double, 6, 1> des_ws_limits_;                                               // Vector6d, filled by operator()
  const double ws[6] = {1.0, -1.0, 0.3, -0.3, 1.1, 0.2};
  for (int i = 0; i < 6; i++) des_ws_limits_(i) = ws[i];
  int best_agent_id = cf_manager_.evaluateAgents(obstacles_, 100.0, 10.0, 0.001, 1.0, des_ws_limits_);
  for (int i = 0; i < (int)cf_manager_.getPredictedPaths().size(); i++) {                  // :341-347
    if (cf_manager_.getPredictedPaths().at(i).size() > 2) {
      const std::vector<Vector3d> &path = cf_manager_.getPredictedPaths().at(i);
      (void)path.back().x();
    }
  }
  cf_manager_.moveRealEEAgent(obstacles_, time_step_, 1, best_agent_id);
  cf_manager_.resetEEAgents(cf_manager_.getNextPosition(), cf_manager_.getNextVelocity(), obstacles_);
  cf_manager_.startPrediction();
  out.data = {cf_manager_.getNextPosition()[0], cf_manager_.getNextPosition()[1], cf_manager_.getNextPosition()[2]};  // :354-356
  const std::vector<Vector3d> traj = cf_manager_.getPlannedTrajectory();
  (void)traj;
  return cf_manager_.getDistFromGoal();
}

void node_style_init(CfManager &cf_manager_, const PositionMsg &p, const std::vector<Obstacle> &obstacles_, int n) {
  cf_manager_.setInitialPosition(Vector3d(p.data.data()));                                 // :366
  const Vector3d goal(0.5, 0.0, 0.7);
  const std::vector<double> k_a(n, 4.0), k_c(n, 0.025), k_r(n, 0.08), k_d(n, 3.0), k_m(n, 0.0), k_rf(1, 0.02);
  cf_manager_.init(goal, 0.01, obstacles_, k_a, k_c, k_r, k_d, k_m, k_rf, 0.2, 0.25, 0.35, 1500, 1);   // :463-471
  Obstacle o(Vector3d(0.1, 0.2, 0.3), 0.05);
  (void)o.getPosition().transpose();
  // the task loader's and the obstacle callback's forms (B/src/panda_bimanual_control.cpp:52, :302-309)
  std::vector<Obstacle> obs;
  Vector3d obst_pos_(0.0, 0.0, 0.0), obst_vel_(0.0, 0.0, 0.0);
  double radius_ = 0.1;
  obs.push_back(Obstacle{obst_pos_, obst_vel_, radius_});
  obs.at(0).setPosition(Vector3d(p.data[0], p.data[1], p.data[2]));
  obs.at(0).setVelocity(obst_vel_);
  (void)cf_manager_.getGoalPosition();                                                     // :513-517
  (void)cf_manager_.getInitialPosition();
}

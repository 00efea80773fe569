// ros_messages.h -- converters between the planner node's wire formats and the facade's types (last
// two rows), for the maintainer who keeps the reference's ROS node around the new planner. Templated on the message
// types so that this header needs no ROS installation (none exists in the build image; the converters are exercised
// with look-alike structs in tests/test_facade.py):
//   bimanual_planning_ros/Position   float64[3] data                          B/msg/Position.msg:1
//   bimanual_planning_ros/Obstacles  Position[] pos, Position[] vel,
//                                    float64[] radius (first M obstacles)     B/msg/Obstacles.msg:1-3
// Node / topic names to keep (B/launch/planning_node.launch:7-15, B/src/panda_bimanual_control.cpp:247-267): node
// `panda_bimanual_control_node` in namespace `/panda_dual/dual_panda_costp_controller`; in: `position` (tick
// trigger), `obstacles`, `contact_wrench`; out: `goals` (the set-point), `goal_distance`, `commanded_path`,
// `predicted_paths`, `controller_params`, `events`. Parameters under `bimanual_planning/*` (planner_node.h:
// TaskParams lists the keys).
#pragma once

#include <vector>

#include "bimanual_planning_ros/obstacle.h"

namespace ghostplanner {
namespace cfplanner {
namespace ros_msgs {

// Position message -> Vector3d (planCallback: Vector3d(p.data.data()), B/src/panda_bimanual_control.cpp:334)
template <class PositionMsg>
inline Vector3d toVector(const PositionMsg &p) { return Vector3d(p.data[0], p.data[1], p.data[2]); }

// Vector3d -> Position message (the published set-point, :353-357)
template <class PositionMsg>
inline PositionMsg toPositionMsg(const Vector3d &v) {
  PositionMsg m;
  m.data[0] = v[0]; m.data[1] = v[1]; m.data[2] = v[2];
  return m;
}

// obstacleCallback, B/src/panda_bimanual_control.cpp:302-309: the stream carries the first msg.radius.size()
// obstacles (the trailing repulsive one is not streamed); positions and velocities are overwritten, radii are not
template <class ObstaclesMsg>
inline void applyObstaclesMsg(const ObstaclesMsg &msg, std::vector<Obstacle> &obstacles) {
  for (size_t i = 0; i < msg.radius.size(); ++i) {
    obstacles.at(i).setPosition(Vector3d(msg.pos[i].data[0], msg.pos[i].data[1], msg.pos[i].data[2]));
    obstacles.at(i).setVelocity(Vector3d(msg.vel[i].data[0], msg.vel[i].data[1], msg.vel[i].data[2]));
  }
}

// dynamic_obstacle_node's publisher side, B/src/dynamic_obstacle_node.cpp:317-328: all obstacles but the last
template <class ObstaclesMsg, class PositionMsg>
inline ObstaclesMsg toObstaclesMsg(const std::vector<Obstacle> &obstacles) {
  ObstaclesMsg m;
  for (size_t i = 0; i + 1 < obstacles.size(); ++i) {
    m.pos.push_back(toPositionMsg<PositionMsg>(obstacles[i].getPosition()));
    m.vel.push_back(toPositionMsg<PositionMsg>(obstacles[i].getVelocity()));
    m.radius.push_back(obstacles[i].getRadius());
  }
  return m;
}

}  // namespace ros_msgs
}  // namespace cfplanner
}  // namespace ghostplanner

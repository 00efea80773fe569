// obstacle.h -- ghostplanner::cfplanner::Obstacle, the sphere value type of
// the reference (B/include/bimanual_planning_ros/obstacle.h:16-40): name,
// position, velocity, radius with the same constructors and accessors. The
// reference's updateVrepObstacles (CoppeliaSim bridge, B/src/obstacle.cpp:7-11)
// is simulator I/O and not part of this build.
//
// Vector3d is Eigen::Vector3d when PMAF_USE_EIGEN is defined, otherwise the
// minimal Vec3 below, which offers the accessors the planner surface uses
// (x() y() z(), operator[] / operator(), three-double constructor, data()).
#pragma once

#include <string>

#ifdef PMAF_USE_EIGEN
#include "eigen3/Eigen/Dense"
#endif

namespace ghostplanner {
namespace cfplanner {

#ifdef PMAF_USE_EIGEN
using Vector3d = Eigen::Vector3d;
#else
struct Vec3 {
  double v[3];
  Vec3() : v{0.0, 0.0, 0.0} {}
  Vec3(double x, double y, double z) : v{x, y, z} {}
  explicit Vec3(const double *p) : v{p[0], p[1], p[2]} {}
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  const double *data() const { return v; }
  double *data() { return v; }
};
using Vector3d = Vec3;
#endif

class Obstacle {
 private:
  std::string name_;
  Vector3d pos_;
  Vector3d vel_;
  double rad_;

 public:
  Obstacle(const std::string name, const Vector3d pos, const Vector3d vel, const double rad)
      : name_{name}, pos_{pos}, vel_{vel}, rad_{rad} {}
  Obstacle(const Vector3d pos, const double rad) : name_{""}, pos_{pos}, vel_{0, 0, 0}, rad_{rad} {}
  Obstacle(const Vector3d pos, const Vector3d vel, const double rad) : name_{""}, pos_{pos}, vel_{vel}, rad_{rad} {}
  Obstacle() : name_{""}, pos_{0, 0, 0}, vel_{0, 0, 0}, rad_{0} {}
  std::string getName() const { return name_; }
  Vector3d getPosition() const { return pos_; }
  void setPosition(Vector3d pos) { pos_ = pos; }
  void setVelocity(Vector3d vel) { vel_ = vel; }
  Vector3d getVelocity() const { return vel_; }
  double getRadius() const { return rad_; }
};

}  // namespace cfplanner
}  // namespace ghostplanner

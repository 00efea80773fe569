// consumer_check.cpp -- drives ghostplanner::cfplanner::SetPointConsumer / SetPointHandOver
// (include/bimanual_planning_ros/setpoint_consumer.h) with a recorded set-point sequence; host-only (no GPU).
// tests/test_setpoint_consumer.py compares every line with the oracle's restatement (orc_consumer_*).
//   usage: consumer_check <points.bin> <n_points> <velocity> <sx> <sy> <sz> [double-fill-at k]
// points.bin: n x 3 doubles. Output per point:
//   <k> <cycles> <accepted> <refused> <nan> <too_close> <inconsistent> <updates> <15 state doubles>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bimanual_planning_ros/setpoint_consumer.h"

using namespace ghostplanner::cfplanner;

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  const int n = atoi(argv[2]);
  const double velocity = atof(argv[3]);
  const Vector3d start(atof(argv[4]), atof(argv[5]), atof(argv[6]));
  int double_fill_at = -1;
  if (argc > 8 && !strcmp(argv[7], "double-fill-at")) double_fill_at = atoi(argv[8]);
  std::vector<double> pts((size_t)n * 3);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(pts.data(), sizeof(double), pts.size(), f) != pts.size()) return 3;
  fclose(f);

  // the one-slot hand-over contract on its own: refuse while occupied, taking empties the slot
  SetPointHandOver slot;
  Vector3d got;
  bool ok = !slot.occupied && !slot.take(got);
  ok = ok && slot.offer(Vector3d(1, 0, 0)) && slot.occupied && !slot.offer(Vector3d(2, 0, 0));
  ok = ok && slot.take(got) && got.x() == 1 && !slot.occupied && !slot.take(got);
  ok = ok && slot.offer(Vector3d(3, 0, 0)) && slot.take(got) && got.x() == 3;
  slot.offer(Vector3d(4, 0, 0)); slot.clear();
  ok = ok && !slot.occupied;
  printf("B %d\n", ok ? 1 : 0);

  SetPointConsumer c;
  c.reset(start);
  for (int k = 0; k < n; ++k) {
    const Vector3d p(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]);
    long cycles;
    if (k == double_fill_at) {
      // a planner that does not wait for the position response: the second point is refused (buffer size 1)
      c.fillBuffer(p);
      c.fillBuffer(p);
      cycles = 0;
      while (cycles < 1000000) { c.update(velocity / 0.9); ++cycles; if (c.readyForNextPoint()) break; }
    } else {
      cycles = c.deliver(p, velocity, 200000);
    }
    const SetPointConsumer::Counters &n_ = c.counters();
    const Vector3d cg = c.currentGoal(), lg = c.lastGoal(), ng = c.nominalGoal(), ig = c.instantaneousGoal();
    printf("%d %ld %ld %ld %ld %ld %ld %ld %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n",
           k, cycles, n_.accepted, n_.refused, n_.nan, n_.too_close, n_.inconsistent, n_.updates, c.vGoal(), c.vAct(),
           c.nextNg(), cg[0], cg[1], cg[2], lg[0], lg[1], lg[2], ng[0], ng[1], ng[2], ig[0], ig[1], ig[2]);
  }
  return 0;
}

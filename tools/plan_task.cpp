// plan_task.cpp -- head-less run of a task file through the planner node
// mirror (include/bimanual_planning_ros/planner_node.h): the controller is
// replaced by "set-point reached" echo, obstacles move like
// dynamic_obstacle_node. Prints one line per tick:
//   <tick> <best index> <x> <y> <z> <goal distance>
// With --consumer every set-point is handed to the controller-side SetPointConsumer (setpoint_consumer.h: the
// reference's TrajectoryBuffer + followTrajectory acceptance logic, cycled at 1 kHz until it asks for the next
// point, like VrepController::targetPoseCallback) and each tick prints a second line
//   C <tick> <controller cycles> <v_goal> <next_ng> <accepted> <refused> <nan> <too_close> <inconsistent>
// --lag F (closed loop, task files with open_loop: false): the controller does not reach the set-point -- the position it
//   reports back is  set-point - F * (set-point - previous reported position)  (a deterministic tracking error), which
//   planCallback hands to CfManager::setRealEEAgentPosition in front of the tick (B/src/panda_bimanual_control.cpp:333-335)
// --goal-ticks N: leave every plan goal after N ticks even if it is not reached (a goal change mid-run: the best agent of
//   the running population is carried into the next init, B/src/cf_manager.cpp:344-354)
// --random-vecs f.bin: explicit Random-agent vectors [N][n_obs][3]; a file with several such blocks gives the k-th plan
//   goal's init() the k-th block (the reference draws fresh vectors in every init)
// usage: plan_task <task.yaml> --start x y z [--max-ticks N] [--goal-ticks N] [--lag F] [--seed S] [--random-vecs f.bin]
//                  [--consumer] [--dump-params]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bimanual_planning_ros/planner_node.h"

using namespace ghostplanner::cfplanner;

static void dump(const TaskParams &t) {
  printf("{\"num_agents_ee\": %d, \"num_agents_body\": %d, \"k_attr\": %.17g, \"k_circ\": %.17g, \"k_repel\": %.17g, "
         "\"k_damp\": %.17g, \"k_manip\": %.17g, \"k_repel_body\": %.17g, \"k_goal_dist\": %.17g, \"k_path_len\": %.17g, "
         "\"k_safe_dist\": %.17g, \"k_workspace\": %.17g, \"max_prediction_steps\": %d, \"prediction_freq_multiple\": %d, "
         "\"approach_dist\": %.17g, \"detect_shell_rad\": %.17g, \"frequency_ros\": %.17g, \"velocity\": %.17g, "
         "\"open_loop\": %s, \"desired_ws_limits\": [",
         t.num_agents_ee, t.num_agents_body, t.k_attr, t.k_circ, t.k_repel, t.k_damp, t.k_manip, t.k_repel_body,
         t.k_goal_dist, t.k_path_len, t.k_safe_dist, t.k_workspace, t.max_prediction_steps, t.prediction_freq_multiple,
         t.approach_dist, t.detect_shell_rad, t.frequency_ros, t.velocity, t.open_loop ? "true" : "false");
  for (int i = 0; i < 6; ++i) printf("%s%.17g", i ? ", " : "", t.desired_ws_limits(i));
  printf("], \"obstacles\": [");
  for (size_t i = 0; i < t.obstacles.size(); ++i) {
    Vector3d p = t.obstacles[i].getPosition(), v = t.obstacles[i].getVelocity();
    printf("%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g]", i ? ", " : "", p[0], p[1], p[2], v[0], v[1], v[2],
           t.obstacles[i].getRadius());
  }
  printf("], \"goals\": [");
  for (size_t i = 0; i < t.goals.size(); ++i) {
    const GoalSpec &g = t.goals[i];
    printf("%s{\"type\": \"%s\", \"end_condition\": \"%s\", \"pos\": [%.17g, %.17g, %.17g], \"overrides\": {", i ? ", " : "",
           g.type.c_str(), g.end_condition.c_str(), g.pos[0], g.pos[1], g.pos[2]);
    bool first = true;
    for (auto &kv : g.overrides) { printf("%s\"%s\": %.17g", first ? "" : ", ", kv.first.c_str(), kv.second); first = false; }
    printf("}}");
  }
  printf("]}\n");
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: plan_task <task.yaml> --start x y z [--max-ticks N] [--seed S] [--dump-params]\n"); return 2; }
  double start[3] = {0, 0, 0};
  long max_ticks = 5000, goal_ticks = -1;
  double lag = 0.0;
  unsigned long long seed = 1;
  bool dump_only = false, with_consumer = false;
  const char *rv_file = nullptr;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "--start") && i + 3 < argc) { for (int c = 0; c < 3; ++c) start[c] = atof(argv[++i]); }
    else if (!strcmp(argv[i], "--max-ticks") && i + 1 < argc) max_ticks = atol(argv[++i]);
    else if (!strcmp(argv[i], "--goal-ticks") && i + 1 < argc) goal_ticks = atol(argv[++i]);
    else if (!strcmp(argv[i], "--lag") && i + 1 < argc) lag = atof(argv[++i]);
    else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 10);
    else if (!strcmp(argv[i], "--dump-params")) dump_only = true;
    else if (!strcmp(argv[i], "--consumer")) with_consumer = true;
    else if (!strcmp(argv[i], "--random-vecs") && i + 1 < argc) rv_file = argv[++i];
  }
  try {
    TaskParams task = loadTaskFile(argv[1]);
    if (dump_only) { dump(task); return 0; }
    PlannerNode node(task, seed);
    std::vector<double> rv_all;   // explicit Random-agent vectors, one [N][n_obs][3] block per plan goal (tests)
    const size_t rv_block = (size_t)task.num_agents_ee * task.obstacles.size() * 3;
    if (rv_file) {
      FILE *f = fopen(rv_file, "rb");
      if (!f) throw std::runtime_error("cannot read --random-vecs file");
      std::vector<double> blk(rv_block);
      while (fread(blk.data(), sizeof(double), rv_block, f) == rv_block) rv_all.insert(rv_all.end(), blk.begin(), blk.end());
      fclose(f);
      if (rv_all.empty()) throw std::runtime_error("--random-vecs file holds no complete [N][n_obs][3] block");
    }
    size_t plan_goal = 0;
    DynamicObstacleSource source(task.obstacles, task.frequency_ros);
    Position position{{start[0], start[1], start[2]}};
    node.planCallback(position, nullptr);              // planning not active: records the initial position
    long tick = 0;
    for (const GoalSpec &goal : task.goals) {
      if (goal.type != "plan") continue;               // key / gesture / goto ...: operator or robot actions
      if (!rv_all.empty()) {
        const size_t k = std::min(plan_goal, rv_all.size() / rv_block - 1);
        node.manager().setRandomVectors(std::vector<double>(rv_all.begin() + k * rv_block, rv_all.begin() + (k + 1) * rv_block));
      }
      if (plan_goal > 0) node.planCallback(position, nullptr);   // between goals the controller's position keeps arriving (:364-367)
      ++plan_goal;
      const long goal_start = tick;
      Position sp = node.startPlan(goal);
      SetPointConsumer consumer;                         // the controller is reset at the start pose ...
      consumer.reset(Vector3d(start[0], start[1], start[2]));
      if (with_consumer)                                 // ... and receives the first published point (:514-518)
        consumer.deliver(Vector3d(sp.data[0], sp.data[1], sp.data[2]), node.velocity());
      position = Position{{sp.data[0], sp.data[1], sp.data[2] - 0.00001}};
      Vector3d prev(position.data[0], position.data[1], position.data[2]);
      while (tick < max_ticks) {
        int best = -1;
        Position next;
        node.planCallback(position, &next, &best);     // controller is ready for the next set-point
        Vector3d nv(next.data[0], next.data[1], next.data[2]);
        std::string why;
        if (!validateSetPoint(prev, nv, &why)) fprintf(stderr, "tick %ld: rejected by the consumer contract: %s\n", tick, why.c_str());
        printf("%ld %d %.17g %.17g %.17g %.17g\n", tick, best, next.data[0], next.data[1], next.data[2], node.goalDistance());
        if (with_consumer) {
          const long cycles = consumer.deliver(nv, node.velocity());
          const SetPointConsumer::Counters &cn = consumer.counters();
          printf("C %ld %ld %.17g %.17g %ld %ld %ld %ld %ld\n", tick, cycles, consumer.vGoal(), consumer.nextNg(),
                 cn.accepted, cn.refused, cn.nan, cn.too_close, cn.inconsistent);
        }
        prev = nv;
        if (lag != 0.0)                                // the controller trails the set-point (closed loop, --lag)
          for (int c = 0; c < 3; ++c) position.data[c] = next.data[c] - lag * (next.data[c] - position.data[c]);
        else
          position = next;                             // echo: the set-point is reached
        node.obstacleCallback(source.step());          // obstacle stream between ticks
        ++tick;
        if (goal.end_condition == "reached" && node.reached()) break;
        if (goal_ticks >= 0 && tick - goal_start >= goal_ticks) break;
      }
      node.finishGoal();
      printf("# goal %s after %ld ticks, distance %.6g, planned trajectory %zu points\n", node.reached() ? "reached" : "not reached", tick,
             node.goalDistance(), node.manager().getPlannedTrajectory().size());
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "plan_task: %s\n", e.what());
    return 1;
  }
  return 0;
}

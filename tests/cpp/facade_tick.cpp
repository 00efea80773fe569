// facade_tick.cpp -- drives the C++ facade (include/bimanual_planning_ros/
// cf_manager.h) exactly like the reference's planner node does
// (B/src/panda_bimanual_control.cpp:463-471, 501-510, 329-369) on the static1
// task scene and prints, per tick, best index / type and the next set-point.
// tests/test_facade.py compares the output with the oracle.
//   usage: facade_tick <n_agents> <max_prediction_steps> <n_ticks> <random_vecs.bin> [viz | comm]   |   facade_tick health
// With `viz` the node's visualize_predicted_paths loop (B/src/panda_bimanual_control.cpp:340-347: 3 N + 1
// getPredictedPaths() calls per tick) runs in every tick and the last lines report the median tick time with and
// without it ("V <us with viz> <us without> <path points visited>"); it also exercises the move operations.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "bimanual_planning_ros/cf_manager.h"

using namespace ghostplanner::cfplanner;

// `health` mode (ABI 5): failure detection and the selected path through the facade. One Had-heuristic agent (the
// reference's layout puts it first, B/src/cf_manager.cpp:70-104) flies straight at an obstacle whose centre lies on the
// agent-goal line: at first contact the REAL agent's rotation vector is 0 / 0 (B/src/cf_agent.cpp:599-611), its force and
// set-point turn NaN: by default planTick hands the NaN on like the reference (health word set), with
// setThrowOnNumericFault(true) it throws at that very tick. Until then every planTick's selected path
// (pmaf_view_winner_path) must be the best agent's predicted path as it was scored.
static int health_mode() {
  std::vector<Obstacle> obstacles = {Obstacle(Vector3d(0.0, 0.0, 0.7), Vector3d(0, 0, 0), 0.05),
                                     Obstacle(Vector3d(100.0, 100.0, 100.0), Vector3d(0, 0, 0), 0.1)};
  const Vector3d start(-0.6, 0.0, 0.7), goal(0.6, 0.0, 0.7);
  Vector6d ws;
  const double wsv[6] = {1.0, -1.0, 0.3, -0.3, 1.1, 0.2};
  for (int i = 0; i < 6; ++i) ws(i) = wsv[i];
  // DEFAULT behaviour = the reference's: the NaN set-point is published, nothing throws (its consumer logs it,
  // B/src/costp_controller.cpp:317-319); the health word reports it
  int t_default = -1;
  {
    CfManager d;
    d.setInitialPosition(start);
    d.init(goal, 0.01, obstacles, {4.0}, {0.025}, {0.08}, {3.0}, {0.0}, {0.02}, 0.2, 0.25, 0.35, 61, 1);
    d.setInitialPosition(start);
    for (int t = 0; t < 400 && t_default < 0; ++t) {
      Vector3d np(0, 0, 0);
      try { d.planTick(obstacles, 0.01, 100.0, 10.0, 0.001, 1.0, ws, &np); }
      catch (const std::exception &e) { fprintf(stderr, "planTick threw without the opt-in: %s\n", e.what()); return 10; }
      const bool nan = np.x() != np.x() || np.y() != np.y() || np.z() != np.z();
      const bool bits = (d.getHealth() & (PMAF_HEALTH_FORCE_NAN | PMAF_HEALTH_SETPOINT_NAN)) != 0;
      if (nan != bits) { fprintf(stderr, "health word and set-point disagree at tick %d\n", t); return 11; }
      if (nan) t_default = t;
    }
    if (t_default < 0) { fprintf(stderr, "the real agent never met the degenerate obstacle (default mode)\n"); return 9; }
  }
  CfManager m;
  m.setThrowOnNumericFault(true);   // opt-in: the NaN as an exception
  m.setInitialPosition(start);
  m.init(goal, 0.01, obstacles, {4.0}, {0.025}, {0.08}, {3.0}, {0.0}, {0.02}, 0.2, 0.25, 0.35, 61, 1);
  m.setInitialPosition(start);
  m.enableSelectedPath();
  long checked = 0;
  for (int t = 0; t < 400; ++t) {
    m.stopPrediction();
    const std::vector<std::vector<Vector3d>> scored = m.getPredictedPaths();
    int best = -1;
    try {
      best = m.planTick(obstacles, 0.01, 100.0, 10.0, 0.001, 1.0, ws);
    } catch (const std::runtime_error &e) {
      const int hb = m.getHealth();
      if (!(hb & PMAF_HEALTH_FORCE_NAN) || !(hb & PMAF_HEALTH_SETPOINT_NAN)) { fprintf(stderr, "throw without health bits: %d\n", hb); return 7; }
      if (t != t_default) { fprintf(stderr, "throw at tick %d, NaN without the opt-in at tick %d\n", t, t_default); return 12; }
      printf("H %d %d\nS %ld\n", t, hb, checked);
      return 0;
    }
    if (m.getHealth() & (PMAF_HEALTH_FORCE_NAN | PMAF_HEALTH_SETPOINT_NAN)) { fprintf(stderr, "health bits without a throw\n"); return 8; }
    int idx = -1;
    const std::vector<Vector3d> sel = m.getSelectedPath(&idx);
    bool ok = idx == best && sel.size() == scored[best].size();
    // (bit patterns: the predicted agent meets the degenerate obstacle some ticks before the real one, its path then
    // carries NaNs -- and must carry the same ones in both views)
    for (size_t k = 0; ok && k < sel.size(); ++k) {
      const double u[3] = {sel[k].x(), sel[k].y(), sel[k].z()}, w[3] = {scored[best][k].x(), scored[best][k].y(), scored[best][k].z()};
      ok = std::memcmp(u, w, sizeof(u)) == 0;
    }
    if (!ok) { fprintf(stderr, "selected path mismatch at tick %d\n", t); return 6; }
    ++checked;
  }
  fprintf(stderr, "the real agent never met the degenerate obstacle\n");
  return 9;
}

// `lat` mode: what one planCallback costs at the C++ boundary on an idle stream (the previous rollout has finished, as in a
// 100 Hz loop), no interpreter in the way: planTick open loop, setRealEEAgentPosition + planTick closed loop
// (B/src/panda_bimanual_control.cpp:333-335: the measured position trails the set-point), and the node's five individual
// calls. usage: facade_tick lat <n_agents> <max_prediction_steps> <n_samples>
static int lat_mode(int N, int cap, int n) {
  std::vector<Obstacle> obstacles;
  const double xs[3] = {0.125, 0.125, -0.35}, zs[3] = {1.0, 0.7, 0.6}, ys[3] = {0.0, 0.125, -0.125};
  for (int g = 0; g < 3; ++g)
    for (int k = 0; k < 3; ++k) obstacles.push_back(Obstacle(Vector3d(xs[g], ys[k], zs[g]), Vector3d(0, 0, 0), 0.1));
  obstacles.push_back(Obstacle(Vector3d(100.0, 100.0, 100.0), Vector3d(0, 0, 0), 0.1));
  const Vector3d start(-0.6, 0.0, 0.75), goal(0.5, 0.0, 0.7);
  Vector6d ws;
  const double wsv[6] = {1.0, -1.0, 0.3, -0.3, 1.1, 0.2};
  for (int i = 0; i < 6; ++i) ws(i) = wsv[i];
  auto stats = [](std::vector<double> v, const char *what) {
    std::sort(v.begin(), v.end());
    printf("L %-44s median %7.2f  p90 %7.2f  p99 %7.2f  max %7.2f us  (%zu samples)\n", what, v[v.size() / 2], v[v.size() * 9 / 10],
           v[v.size() * 99 / 100], v.back(), v.size());
  };
  for (int mode = 0; mode < 3; ++mode) {
    CfManager m;
    m.setRandomSeed(7);
    m.setInitialPosition(start);
    m.init(goal, 0.01, obstacles, std::vector<double>(N, 4.0), std::vector<double>(N, 0.025), std::vector<double>(N, 0.08),
           std::vector<double>(N, 3.0), std::vector<double>(N, 0.0), std::vector<double>(1, 0.02), 0.2, 0.25, 0.35, cap, 1);
    m.setInitialPosition(start);
    Vector3d measured = start;
    std::vector<double> us;
    for (int t = 0; t < n + 20; ++t) {
      if (t % 100 == 0) { m.setInitialPosition(start); measured = start; }   // stay near the start: full-length rollouts
      m.stopPrediction();                                                   // idle stream
      const auto t0 = std::chrono::steady_clock::now();
      Vector3d next;
      if (mode == 2) {
        m.stopPrediction();
        const int best = m.evaluateAgents(obstacles, 100.0, 10.0, 0.001, 1.0, ws);
        m.moveRealEEAgent(obstacles, 0.01, 1, best);
        m.resetEEAgents(m.getNextPosition(), m.getNextVelocity(), obstacles);
        m.startPrediction();
        next = m.getNextPosition();
      } else {
        if (mode == 1) m.setRealEEAgentPosition(measured);
        m.planTick(obstacles, 0.01, 100.0, 10.0, 0.001, 1.0, ws, &next);
      }
      const double dt_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (t >= 20) us.push_back(dt_us);
      for (int c = 0; c < 3; ++c) measured[c] = next[c] - 0.3 * (next[c] - measured[c]);
    }
    stats(us, mode == 0 ? "planTick, open loop" : mode == 1 ? "setRealEEAgentPosition + planTick, closed loop" : "the node's five calls (stop ... start)");
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "health")) return health_mode();
  if (argc > 4 && !strcmp(argv[1], "lat")) return lat_mode(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
  if (argc < 5) return 2;
  const int N = atoi(argv[1]), cap = atoi(argv[2]), ticks = atoi(argv[3]);
  // static1 scene: 9 spheres + repulsive sentinel (values as in pmaf scenes.static1_obstacles)
  std::vector<Obstacle> obstacles;
  const double xs[3] = {0.125, 0.125, -0.35}, zs[3] = {1.0, 0.7, 0.6}, ys[3] = {0.0, 0.125, -0.125};
  for (int g = 0; g < 3; ++g)
    for (int k = 0; k < 3; ++k) obstacles.push_back(Obstacle(Vector3d(xs[g], ys[k], zs[g]), Vector3d(0, 0, 0), 0.1));
  obstacles.push_back(Obstacle(Vector3d(100.0, 100.0, 100.0), Vector3d(0, 0, 0), 0.1));
  std::vector<double> rv((size_t)N * obstacles.size() * 3);
  FILE *f = fopen(argv[4], "rb");
  if (!f || fread(rv.data(), sizeof(double), rv.size(), f) != rv.size()) return 3;
  fclose(f);

  const Vector3d start(-0.6, 0.0, 0.75), goal(0.5, 0.0, 0.7);
  const double dt = 0.01;
  Vector6d ws;
  const double wsv[6] = {1.0, -1.0, 0.3, -0.3, 1.1, 0.2};
  for (int i = 0; i < 6; ++i) ws(i) = wsv[i];

  const bool viz = argc > 5 && !strcmp(argv[5], "viz");
  // `comm`: a one-rank RCCL communicator created through the C-ABI is attached to the manager; after every
  // evaluateAgents the gathered winner record must hold the best agent's index and the path that was scored
  const bool with_comm = argc > 5 && !strcmp(argv[5], "comm");
  pmaf_comm *comm = nullptr;
  long comm_checked = 0;
  CfManager moved_from;
  moved_from.setInitialPosition(start);   // planCallback while planning is not active
  CfManager cf_manager_ = std::move(moved_from);   // cf_manager.h:53 (move construction keeps the state)
  {
    CfManager tmp;
    tmp = std::move(cf_manager_);                  // :55 (move assignment), and back
    cf_manager_ = std::move(tmp);
  }
  cf_manager_.setRandomVectors(rv);
  if (with_comm) {
    unsigned char id[PMAF_COMM_ID_BYTES];
    if (pmaf_comm_unique_id(id) != PMAF_OK || pmaf_comm_init_rccl(1, 0, id, -1, &comm) != PMAF_OK) {
      fprintf(stderr, "communicator: %s\n", pmaf_last_error());
      return 4;
    }
    cf_manager_.attachCommunicator(comm);   // before init(): the attachment survives (re-)initialisation
  }
  auto do_init = [&] {
    cf_manager_.init(goal, dt, obstacles, std::vector<double>(N, 4.0), std::vector<double>(N, 0.025),
                     std::vector<double>(N, 0.08), std::vector<double>(N, 3.0), std::vector<double>(N, 0.0),
                     std::vector<double>(1, 0.02), 0.2, 0.25, 0.35, cap, 1);
  };
  do_init();                               // node start-up, :463-471
  Vector3d current_pos = start;
  do_init();                               // taskCallback PLAN, :501-509
  cf_manager_.setInitialPosition(current_pos);
  std::vector<double> us_viz, us_plain;
  size_t visited = 0;
  double sink = 0.0;
  auto visualize_predicted_path = [&](const std::vector<Vector3d> &poses, int agent, int best_agent) {  // :390-427
    (void)agent; (void)best_agent;
    for (const auto &pose : poses) { sink += pose.x() + pose.y() + pose.z(); ++visited; }
  };
  for (int t = 0; t < ticks; ++t) {        // planCallback, :336-352
    const bool with_viz = viz && (t % 2 == 0);
    const auto t0 = std::chrono::steady_clock::now();
    cf_manager_.stopPrediction();
    std::vector<std::vector<Vector3d>> scored;
    if (with_comm) scored = cf_manager_.getPredictedPaths();   // the paths this evaluation scores
    int best = cf_manager_.evaluateAgents(obstacles, 100.0, 10.0, 0.001, 1.0, ws);
    if (with_comm) {
      const std::vector<double> w = cf_manager_.gatherWinners();
      const size_t rec = cf_manager_.winnerRecordDoubles();
      bool ok = w.size() == rec && (int)w[1] == best && (size_t)w[2] == scored[best].size();
      for (size_t k = 0; ok && k < scored[best].size(); ++k)
        ok = w[8 + 3 * k] == scored[best][k].x() && w[8 + 3 * k + 1] == scored[best][k].y() && w[8 + 3 * k + 2] == scored[best][k].z();
      if (!ok) { fprintf(stderr, "winner record mismatch at tick %d\n", t); return 5; }
      ++comm_checked;
    }
    if (with_viz) {                          // :340-347, verbatim call pattern
      for (int i = 0; i < (int)cf_manager_.getPredictedPaths().size(); i++) {
        if (cf_manager_.getPredictedPaths().at(i).size() > 2) {
          visualize_predicted_path(cf_manager_.getPredictedPaths().at(i), i, best);
        }
      }
    }
    cf_manager_.moveRealEEAgent(obstacles, dt, 1, best);
    cf_manager_.resetEEAgents(cf_manager_.getNextPosition(), cf_manager_.getNextVelocity(), obstacles);
    cf_manager_.startPrediction();
    Vector3d np = cf_manager_.getNextPosition();
    if (viz) {
      cf_manager_.stopPrediction();          // whole tick incl. its rollout, so the two variants are comparable
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      (with_viz ? us_viz : us_plain).push_back(us);
    }
    printf("%d %d %d %.17g %.17g %.17g %.17g\n", t, best, cf_manager_.getBestAgentType(), np.x(), np.y(), np.z(),
           cf_manager_.getDistFromGoal());
  }
  cf_manager_.stopPrediction();
  auto paths = cf_manager_.getPredictedPaths();
  auto lens = cf_manager_.getPredictedPathLengths();
  for (size_t a = 0; a < paths.size(); ++a)
    printf("P %zu %zu %.17g %.17g %.17g %.17g\n", a, paths[a].size(), paths[a].back().x(), paths[a].back().y(),
           paths[a].back().z(), lens[a]);
  printf("T %zu\n", cf_manager_.getPlannedTrajectory().size());
  if (with_comm) {
    printf("W %ld\n", comm_checked);
    cf_manager_.attachCommunicator(nullptr);
    pmaf_comm_destroy(comm);
  }
  if (viz && !us_viz.empty() && !us_plain.empty()) {
    std::sort(us_viz.begin(), us_viz.end());
    std::sort(us_plain.begin(), us_plain.end());
    printf("V %.1f %.1f %zu %g\n", us_viz[us_viz.size() / 2], us_plain[us_plain.size() / 2], visited, sink);
  }
  return 0;
}

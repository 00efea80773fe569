// pin_harness.cpp -- TEST INFRASTRUCTURE ONLY: drives the REFERENCE's own CfManager (B/src/cf_manager.cpp +
// B/src/cf_agent.cpp, compiled unmodified from where they lie by oracle/pin/make_pin.sh against a REAL Eigen3 and REAL
// dqrobotics headers) through the planner node's call sequence and writes what it computes, every double as a C99 hex
// literal, as JSON. tests/test_reference_pin.py holds the CPU oracle to these files bit for bit -- this is what turns
// "parity unpinned" into "pinned" (DESIGN.md section 2). It cannot be built in the build container of this repo (no
// Eigen3, no dqrobotics there); it is written to be run by anyone who can build the reference:
//     bash oracle/pin/make_pin.sh /path/to/predictive-multi-agent-framework
// Build-owned code: nothing here is taken from the reference; it only CALLS its public surface the way
// PandaBimanualPlanning does (B/src/panda_bimanual_control.cpp:329-369 planCallback, :494-511 taskCallback PLAN).
//
// Determinism:
//  * makeRandomVector() (B/src/helper_functions.cpp:7-13 draws from std::random_device) is defined HERE and hands out the
//    scenario's raw triples in call order (RandomCfAgent's constructor normalises them itself with Eigen,
//    B/include/bimanual_planning_ros/cf_agent.h:338-342); B/src/helper_functions.cpp is not compiled.
//  * the reference's rollouts run in one std::thread per agent and are cut by wall clock; this harness lets every
//    rollout run to its guard (full horizon or goal reached) before it issues the next tick: it polls the agents'
//    atomic running flags and path sizes, stops the prediction, and verifies every agent ended at
//    max_prediction_steps points or inside the goal region (else it resumes the prediction and waits again).
//
// usage: pin_harness <scenario.txt> <out.json>     (scenario format: oracle/pin/make_scenarios.py)
#include <gnu/libc-version.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "bimanual_planning_ros/cf_manager.h"

using Eigen::Vector3d;
using ghostplanner::cfplanner::CfManager;
using ghostplanner::cfplanner::Obstacle;

// ---- the hook that makes Random agents reproducible -----------------------------------------------------------------
static std::vector<Vector3d> g_random_triples;
static size_t g_random_next = 0;
Eigen::Vector3d makeRandomVector() {
  if (g_random_next >= g_random_triples.size())
    throw std::runtime_error("scenario holds too few random triples (n_random) for the makeRandomVector() calls made");
  return g_random_triples[g_random_next++];
}

struct Goal { Vector3d pos; long max_ticks; int until_reached; };
struct Scenario {
  std::string name;
  int n_agents = 0, n_body = 1, dump_paths = 0, dynamic = 0, detail_every = 1;
  size_t max_steps = 0, freq_multiple = 1;
  double dt = 0.01, k_attr = 0, k_circ = 0, k_repel = 0, k_damp = 0, k_manip = 0, k_repel_body = 0;
  double vel_max = 0, approach = 0, shell = 0, cost[4] = {0, 0, 0, 0}, ws[6] = {0, 0, 0, 0, 0, 0}, lag = 0.0;
  int closed_loop = 0;
  Vector3d start{0, 0, 0};
  std::vector<Obstacle> obstacles;
  std::vector<Goal> goals;
};

static double num(std::istream &in) {
  std::string t;
  if (!(in >> t)) throw std::runtime_error("scenario: unexpected end of file");
  char *e = nullptr;
  const double v = std::strtod(t.c_str(), &e);   // decimal or C99 hex literal
  if (!e || *e) throw std::runtime_error("scenario: not a number: " + t);
  return v;
}
static Vector3d vec(std::istream &in) { double x = num(in), y = num(in), z = num(in); return Vector3d(x, y, z); }

static Scenario load(const char *path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error(std::string("cannot open ") + path);
  Scenario s;
  std::string key;
  in >> key;
  if (key != "pmaf-pin-scenario") throw std::runtime_error("not a pmaf-pin-scenario file");
  (void)num(in);
  while (in >> key) {
    if (key == "name") in >> s.name;
    else if (key == "n_agents") s.n_agents = (int)num(in);
    else if (key == "n_body") s.n_body = (int)num(in);
    else if (key == "max_prediction_steps") s.max_steps = (size_t)num(in);
    else if (key == "freq_multiple") s.freq_multiple = (size_t)num(in);
    else if (key == "dt") s.dt = num(in);
    else if (key == "gains") { s.k_attr = num(in); s.k_circ = num(in); s.k_repel = num(in); s.k_damp = num(in); s.k_manip = num(in); }
    else if (key == "k_repel_body") s.k_repel_body = num(in);
    else if (key == "limits") { s.vel_max = num(in); s.approach = num(in); s.shell = num(in); }
    else if (key == "cost") for (double &c : s.cost) c = num(in);
    else if (key == "ws") for (double &w : s.ws) w = num(in);
    else if (key == "start") s.start = vec(in);
    else if (key == "closed_loop") { s.closed_loop = (int)num(in); s.lag = num(in); }
    else if (key == "dynamic") s.dynamic = (int)num(in);
    else if (key == "dump_paths") s.dump_paths = (int)num(in);
    else if (key == "detail_every") s.detail_every = (int)num(in);   // per-agent records on every k-th tick (file size)
    else if (key == "obstacles") {
      const int n = (int)num(in);
      for (int i = 0; i < n; ++i) { Vector3d p = vec(in), v = vec(in); const double r = num(in); s.obstacles.push_back(Obstacle(p, v, r)); }
    } else if (key == "goals") {
      const int n = (int)num(in);
      for (int i = 0; i < n; ++i) { Goal g; g.pos = vec(in); g.max_ticks = (long)num(in); g.until_reached = (int)num(in); s.goals.push_back(g); }
    } else if (key == "random") {
      const int n = (int)num(in);
      for (int i = 0; i < n; ++i) g_random_triples.push_back(vec(in));
    } else throw std::runtime_error("scenario: unknown key " + key);
  }
  if (s.n_agents < 1 || s.obstacles.empty() || s.goals.empty() || s.max_steps < 1) throw std::runtime_error("scenario incomplete");
  return s;
}

static void hx(FILE *f, double v) { fprintf(f, "\"%a\"", v); }
static void hx3(FILE *f, const Vector3d &v) { fputc('[', f); hx(f, v[0]); fputc(',', f); hx(f, v[1]); fputc(',', f); hx(f, v[2]); fputc(']', f); }

// every rollout to its guard (see the header); returns the number of resumptions it needed
static int finish_rollouts(CfManager &cf, const Scenario &s, const Vector3d &goal) {
  using namespace std::chrono_literals;
  int resumed = 0;
  for (;;) {
    std::vector<int> last(s.n_agents, -1);
    int stable = 0;
    std::this_thread::sleep_for(1ms);
    while (stable < 3) {
      bool same = true;
      for (int i = 0; i < s.n_agents; ++i) {
        const int n = cf.getNumPredictionSteps(i);     // a size read only: no path is copied while its thread appends
        if (n != last[i]) { same = false; last[i] = n; }
      }
      stable = same ? stable + 1 : 0;
      std::this_thread::sleep_for(1ms);
    }
    cf.stopPrediction();                               // B/src/cf_manager.cpp:126-140: returns once no agent is running
    bool done = true;
    const std::vector<std::vector<Vector3d>> paths = cf.getPredictedPaths();   // safe now: all threads idle
    for (int i = 0; i < s.n_agents; ++i) {
      const bool full = paths[i].size() >= s.max_steps;
      const bool at_goal = !((goal - paths[i].back()).norm() > 0.1);           // the loop guard, B/src/cf_agent.cpp:310
      if (!full && !at_goal) done = false;
    }
    if (done) return resumed;
    ++resumed;                                         // a thread had not been scheduled yet: let it carry on
    if (resumed > 10000) throw std::runtime_error("rollouts do not finish");
    cf.startPrediction();
  }
}

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: pin_harness <scenario.txt> <out.json>\n"); return 2; }
  try {
    const Scenario s = load(argv[1]);
    FILE *f = fopen(argv[2], "w");
    if (!f) throw std::runtime_error(std::string("cannot write ") + argv[2]);
    // what the numbers depend on besides the sources: Eigen's version and whether its packet path is compiled in (the
    // 3-vector dot-product association, include/pmaf.h pmaf_eval_order), the compiler, the libm behind std::exp
    fprintf(f, "{\"format\": \"pmaf-reference-pin-1\", \"scenario\": \"%s\",\n \"meta\": {\"eigen\": \"%d.%d.%d\", \"eigen_vectorize\": %s, "
               "\"eigen_dont_vectorize\": %s, \"compiler\": \"%s\", \"optimize\": %s, \"fma_contraction_possible\": %s, \"glibc\": \"%s\", \"cpu_fma\": %s},\n",
            s.name.c_str(), EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION,
#ifdef EIGEN_VECTORIZE
            "true",
#else
            "false",
#endif
#ifdef EIGEN_DONT_VECTORIZE
            "true",
#else
            "false",
#endif
            __VERSION__,
#ifdef __OPTIMIZE__
            "true",
#else
            "false",
#endif
#ifdef __FMA__
            "true",
#else
            "false",
#endif
            gnu_get_libc_version(),
            // glibc >= 2.28 picks its exp variant by the CPU at run time (ifunc), not by this file's compile flags: with FMA
            // it is the algorithm the kernels and the oracle's portable mode restate (tools/gen_exp_table.py)
#if defined(__x86_64__)
            __builtin_cpu_supports("fma") ? "true" : "false"
#else
            "null"
#endif
            );
    std::vector<Obstacle> obstacles = s.obstacles;
    const Eigen::Matrix<double, 6, 1> ws = (Eigen::Matrix<double, 6, 1>() << s.ws[0], s.ws[1], s.ws[2], s.ws[3], s.ws[4], s.ws[5]).finished();
    const int n = s.n_agents;
    auto init = [&](CfManager &cf, const Vector3d &goal) {   // B/src/panda_bimanual_control.cpp:463-471 / :501-509
      cf.init(goal, s.dt, obstacles, std::vector<double>(n, s.k_attr), std::vector<double>(n, s.k_circ),
              std::vector<double>(n, s.k_repel), std::vector<double>(n, s.k_damp), std::vector<double>(n, s.k_manip),
              std::vector<double>(s.n_body, s.k_repel_body), s.vel_max, s.approach, s.shell, s.max_steps, s.freq_multiple);
    };
    CfManager cf;                                       // the node's member, default-constructed
    Vector3d position = s.start;
    cf.setInitialPosition(position);                    // planCallback while planning is inactive, :364-367
    fprintf(f, " \"goals\": [\n");
    for (size_t gi = 0; gi < s.goals.size(); ++gi) {
      const Goal &g = s.goals[gi];
      // taskCallback, GoalType::PLAN, :494-511
      if (gi > 0) cf.setInitialPosition(position);
      const Vector3d current = cf.getNextPosition();
      const size_t rnd0 = g_random_next;
      init(cf, g.pos);
      cf.setInitialPosition(current);
      const Vector3d ip = cf.getInitialPosition();
      position = Vector3d(ip[0], ip[1], (ip[2] + 0.00001) - 0.00001);   // the first published point, echoed (:514-518)
      fprintf(f, "  {\"goal\": "); hx3(f, g.pos);
      fprintf(f, ", \"start\": "); hx3(f, current);
      fprintf(f, ", \"random_first\": %zu, \"random_used\": %zu, \"ticks\": [\n", rnd0, g_random_next - rnd0);
      long t = 0;
      for (; t < g.max_ticks; ++t) {
        if (s.closed_loop) cf.setRealEEAgentPosition(position);                       // :333-335
        cf.stopPrediction();                                                         // :336
        const int best = cf.evaluateAgents(obstacles, s.cost[0], s.cost[1], s.cost[2], s.cost[3], ws);   // :337-339
        // what that selection scored
        const std::vector<std::vector<Vector3d>> paths = cf.getPredictedPaths();
        const std::vector<double> lens = cf.getPredictedPathLengths();
        const std::vector<bool> ok = cf.getAgentSuccess();
        cf.moveRealEEAgent(obstacles, s.dt, 1, best);                                 // :348
        cf.resetEEAgents(cf.getNextPosition(), cf.getNextVelocity(), obstacles);      // :350-351
        cf.startPrediction();                                                        // :352
        const Vector3d next = cf.getNextPosition();
        fprintf(f, "%s   {\"best\": %d, \"type\": %d, \"pos\": ", t ? ",\n" : "", best, cf.getBestAgentType()); hx3(f, next);
        fprintf(f, ", \"vel\": "); hx3(f, cf.getNextVelocity());
        fprintf(f, ", \"force\": "); hx3(f, cf.getEEForce());
        fprintf(f, ", \"dist\": "); hx(f, cf.getDistFromGoal());
        const bool detail = s.detail_every <= 1 || t % s.detail_every == 0;
        if (detail) {
        fprintf(f, ", \"n\": [");
        for (int i = 0; i < n; ++i) fprintf(f, "%s%zu", i ? "," : "", paths[i].size());
        fprintf(f, "], \"len\": [");
        for (int i = 0; i < n; ++i) { if (i) fputc(',', f); hx(f, lens[i]); }
        fprintf(f, "], \"reached\": [");
        for (int i = 0; i < n; ++i) fprintf(f, "%s%d", i ? "," : "", ok[i] ? 1 : 0);
        fprintf(f, "], \"last\": [");
        for (int i = 0; i < n; ++i) { if (i) fputc(',', f); hx3(f, paths[i].back()); }
        fputc(']', f);
        if (s.dump_paths) {
          fprintf(f, ", \"paths\": [");
          for (int i = 0; i < n; ++i) {
            fprintf(f, "%s[", i ? "," : "");
            for (size_t k = 0; k < paths[i].size(); ++k) { if (k) fputc(',', f); hx3(f, paths[i][k]); }
            fputc(']', f);
          }
          fputc(']', f);
        }
        }
        const int resumed = finish_rollouts(cf, s, g.pos);
        fprintf(f, ", \"resumed\": %d}", resumed);
        position = s.closed_loop ? Vector3d(next - s.lag * (next - position)) : next;   // the controller's report
        if (s.dynamic)                                                                  // dynamic_obstacle_node, :355-357
          for (size_t i = 0; i + 1 < obstacles.size(); ++i)
            obstacles[i].setPosition(obstacles[i].getPosition() + obstacles[i].getVelocity() / 100.0);
        if (g.until_reached && cf.getDistFromGoal() < 0.01) { ++t; break; }             // EndCondition::REACHED, :565-569
      }
      fprintf(f, "\n  ], \"n_ticks\": %ld, \"planned_trajectory\": %zu}%s\n", t, cf.getPlannedTrajectory().size(),
              gi + 1 < s.goals.size() ? "," : "");
    }
    fprintf(f, " ]}\n");
    fclose(f);
    cf.joinPredictionThreads();
  } catch (const std::exception &e) {
    fprintf(stderr, "pin_harness: %s\n", e.what());
    return 1;
  }
  return 0;
}

"""The C++ facade ghostplanner::cfplanner::CfManager / Obstacle
(include/bimanual_planning_ros/) driven like the reference's planner node
(tests/cpp/facade_tick.cpp) must reproduce the oracle."""
import os
import subprocess

import numpy as np
import pytest

import conftest

ROOT = conftest.ROOT


def test_ros_message_converters_with_lookalike_structs(tmp_path):
    """include/bimanual_planning_ros/ros_messages.h against structs shaped like the generated ROS messages
    (Position: boost::array<double,3> data; Obstacles: vectors of Position + radius), host-only"""
    src = r'''
#include <array>
#include <cstdio>
#include <vector>
#include "bimanual_planning_ros/ros_messages.h"
using namespace ghostplanner::cfplanner;
struct Position { std::array<double, 3> data; };
struct Obstacles { std::vector<Position> pos, vel; std::vector<double> radius; };
int main() {
  std::vector<Obstacle> obs = {Obstacle(Vector3d(1, 2, 3), Vector3d(0.1, 0, 0), 0.2), Obstacle(Vector3d(4, 5, 6), Vector3d(0, 0.2, 0), 0.3),
                               Obstacle(Vector3d(100, 100, 100), Vector3d(0, 0, 0), 0.1)};
  Obstacles m = ros_msgs::toObstaclesMsg<Obstacles, Position>(obs);
  if (m.radius.size() != 2 || m.pos[1].data[2] != 6 || m.vel[0].data[0] != 0.1) return 1;   // the last one is not streamed
  m.pos[0].data[0] = 7.5; m.vel[1].data[1] = -0.4; m.radius[0] = 9.0;
  ros_msgs::applyObstaclesMsg(m, obs);
  if (obs[0].getPosition().x() != 7.5 || obs[1].getVelocity().y() != -0.4) return 2;
  if (obs[0].getRadius() != 0.2 || obs[2].getPosition().x() != 100) return 3;              // radii and the sentinel untouched
  Position p = ros_msgs::toPositionMsg<Position>(Vector3d(0.5, -0.25, 0.75));
  Vector3d v = ros_msgs::toVector(p);
  if (v.x() != 0.5 || v.y() != -0.25 || v.z() != 0.75) return 4;
  return 0;
}
'''
    exe = str(tmp_path / "rosmsg")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-x", "c++", "-", "-o", exe],
                       input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert subprocess.run([exe]).returncode == 0


def test_facade_headers_compile_standalone():
    """host-only: the facade is plain C++17 over include/pmaf.h"""
    src = "#include \"bimanual_planning_ros/cf_manager.h\"\nint main(){ghostplanner::cfplanner::CfManager m; (void)m; return 0;}\n"
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-x", "c++", "-"], input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


@pytest.mark.gpu
def test_facade_node_sequence_matches_oracle(pmaf, oracle, scenes, tmp_path, hip_lib):
    oracle.set_exp_mode(1)
    try:
        N, cap, ticks = 12, 151, 20
        sc = scenes.static1_scene(N, cap - 1)
        rvf = tmp_path / "rv.bin"
        np.ascontiguousarray(sc["random_vecs"]).tofile(rvf)
        exe = conftest.exe(os.path.join(ROOT, "tests", "cpp", "facade_tick"))
        out = subprocess.run([exe, str(N), str(cap), str(ticks), str(rvf)], capture_output=True, check=True,
                             env=conftest.binary_env(pmaf)).stdout.decode()
        ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
        ora.set_initial_position(sc["start"])
        lines = out.strip().split("\n")
        for t in range(ticks):
            f = lines[t].split()
            b = ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
            assert int(f[0]) == t and int(f[1]) == b and int(f[2]) == ora.best_type()
            np.testing.assert_array_equal(np.array(f[3:6], dtype=float), ora.real_state()[0])
            assert float(f[6]) == ora.dist_from_goal()
        po, no = ora.paths()
        pl = ora.path_lengths()
        for a in range(N):
            f = lines[ticks + a].split()
            assert f[0] == "P" and int(f[1]) == a and int(f[2]) == no[a]
            np.testing.assert_array_equal(np.array(f[3:6], dtype=float), po[a, no[a] - 1])
            assert float(f[6]) == pl[a]
        assert lines[ticks + N].split() == ["T", str(len(ora.real_path()))]
    finally:
        oracle.set_exp_mode(0)


def test_eigen_branch_api_shape():
    """The PMAF_USE_EIGEN branch of the facade headers through a compiler. API-SHAPE check only:
    tests/cpp/eigen_api_check/eigen3/Eigen/Dense declares -- implements nothing of -- the Eigen::Matrix members the
    facade and a node-style caller use, with Eigen 3.3's documented signatures; `g++ -fsyntax-only` over every facade
    header, the C++ test drivers and the reference's call forms (node_style_caller.cpp). It pins no numerics and is no
    substitute for test_facade_compiles_against_real_eigen_if_present (skipped here: no Eigen3 in the image)."""
    chk = os.path.join(ROOT, "tests", "cpp", "eigen_api_check")
    base = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DPMAF_USE_EIGEN", "-I" + chk,
            "-I" + os.path.join(chk, "eigen3"), "-I" + os.path.join(ROOT, "include")]
    units = [os.path.join(ROOT, "tests", "cpp", "facade_tick.cpp"), os.path.join(ROOT, "tests", "cpp", "consumer_check.cpp"),
             os.path.join(ROOT, "tools", "plan_task.cpp"), os.path.join(chk, "node_style_caller.cpp")]
    for u in units:
        r = subprocess.run(base + [u], capture_output=True)
        assert r.returncode == 0, u + "\n" + r.stderr.decode()
    for hdr in ("cf_manager.h", "obstacle.h", "planner_node.h", "setpoint_consumer.h", "ros_messages.h"):
        r = subprocess.run(base + ["-x", "c++", "-"], input=('#include "bimanual_planning_ros/%s"\n' % hdr).encode(),
                           capture_output=True)
        assert r.returncode == 0, hdr + "\n" + r.stderr.decode()
    # the check bites: a call form Eigen does not offer (a std::vector where the 6-vector is expected) is rejected
    bad = '#include "bimanual_planning_ros/cf_manager.h"\nint f(ghostplanner::cfplanner::CfManager &m, std::vector<ghostplanner::cfplanner::Obstacle> &o) {\n' \
          '  return m.evaluateAgents(o, 1, 1, 1, 1, std::vector<double>(6, 0.0)); }\n'
    r = subprocess.run(base + ["-x", "c++", "-"], input=bad.encode(), capture_output=True)
    assert r.returncode != 0


def test_facade_compiles_against_real_eigen_if_present():
    """the PMAF_USE_EIGEN branch (a ROS box): compile-only, skipped where no
    Eigen3 is installed (this image has none)"""
    import glob
    cands = [d for d in ("/usr/include/eigen3", "/usr/local/include/eigen3") if os.path.exists(os.path.join(d, "Eigen", "Dense"))]
    cands += [os.path.dirname(os.path.dirname(p)) for p in glob.glob("/opt/*/include/eigen3/Eigen/Dense")]
    if not cands:
        pytest.skip("no Eigen3 headers in this image")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DPMAF_USE_EIGEN", "-I" + cands[0],
                        "-I" + os.path.dirname(cands[0]), "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "facade_tick.cpp")], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


@pytest.mark.gpu
def test_node_visualisation_loop_does_not_dominate_the_tick(scenes, tmp_path, hip_lib):
    """the reference node with visualize_predicted_paths: true (every shipped task)
    calls getPredictedPaths() 3 N + 1 times per tick (panda_bimanual_control.cpp:340-347).
    As shipped: N = 10 agents, max_prediction_steps 1500. A tick with the loop must
    cost less than twice a tick without it (one D2H per rollout + cached conversion)."""
    N, cap, ticks = 10, 1500, 60
    sc = scenes.static1_scene(N, cap - 1)
    rvf = tmp_path / "rv.bin"
    np.ascontiguousarray(sc["random_vecs"]).tofile(rvf)
    exe = conftest.exe(os.path.join(ROOT, "tests", "cpp", "facade_tick"))
    out = subprocess.run([exe, str(N), str(cap), str(ticks), str(rvf), "viz"], capture_output=True, check=True).stdout.decode()
    v = [l for l in out.strip().split("\n") if l.startswith("V ")][0].split()
    with_viz, plain, visited = float(v[1]), float(v[2]), int(v[3])
    print("tick with the visualisation loop %.0f us, without %.0f us (%d path points visited)" % (with_viz, plain, visited))
    assert visited > 1000 * ticks // 2
    assert with_viz < 2.0 * plain


@pytest.mark.gpu
def test_facade_with_an_attached_rccl_communicator(scenes, tmp_path, hip_lib):
    """C++ host, as INTEGRATION.md section 4 shows it: pmaf_comm_unique_id / pmaf_comm_init_rccl (one rank),
    CfManager::attachCommunicator before init(), the node's five-call sequence; after every evaluateAgents the
    gathered winner record holds the best index and the scored path"""
    N, cap, ticks = 12, 151, 25
    sc = scenes.static1_scene(N, cap - 1)
    rvf = tmp_path / "rv.bin"
    np.ascontiguousarray(sc["random_vecs"]).tofile(rvf)
    exe = conftest.exe(os.path.join(ROOT, "tests", "cpp", "facade_tick"))
    r = subprocess.run([exe, str(N), str(cap), str(ticks), str(rvf), "comm"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert ("W %d" % ticks) in r.stdout.decode()

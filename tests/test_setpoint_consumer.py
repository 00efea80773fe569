"""the set-point consumer contract. The C++ class
ghostplanner::cfplanner::SetPointConsumer (include/bimanual_planning_ros/
setpoint_consumer.h: TrajectoryBuffer + the trajectory half of
CoSTPController::followTrajectory) against the oracle's restatement of the
same reference code (orc_consumer_*), fed with the set-point sequences the
planner emits on the shipped static1 / dyn1 tasks and with the edge cases the
reference handles (NaN, points < 1e-6 m apart, a refused second point,
inconsistent trajectories). Host-only: no GPU involved."""
import os
import subprocess

import numpy as np
import pytest

import conftest

ROOT = conftest.ROOT


@pytest.fixture(scope="module")
def consumer_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("consumer") / "consumer_check")
    # (the consumer's 3-vector arithmetic follows the evaluation-order policy of the pair under test)
    defs = ["-DPMAF_DOT_RIGHT_ASSOC"] if os.environ.get("PMAF_VARIANT") == "rassoc" else []
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-ffp-contract=off"] + defs +
                          ["-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "consumer_check.cpp"), "-o", exe])
    return exe


def planner_set_points(oracle, scenes, name, max_ticks):
    """the set-point sequence of a head-less task run (oracle planner, moving obstacles)"""
    import json
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "task_scenes.json")))[name]
    sc = scenes.scene_from_record(rec, name)
    o = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
    o.set_initial_position(sc["start"])
    obs = sc["obstacles"].copy()
    pts = []
    for t in range(max_ticks):
        o.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        obs = scenes.advance_live_obstacles(obs)
        pts.append(o.real_state()[0].copy())
        if o.dist_from_goal() < 0.01:
            break
    o.close()
    return sc, np.asarray(pts)


def run_both(exe, oracle, tmp_path, pts, start, velocity, double_fill_at=-1):
    f = tmp_path / "pts.bin"
    np.ascontiguousarray(pts, dtype=np.float64).tofile(f)
    cmd = [exe, str(f), str(len(pts)), repr(float(velocity))] + [repr(float(x)) for x in start]
    if double_fill_at >= 0:
        cmd += ["double-fill-at", str(double_fill_at)]
    out = subprocess.run(cmd, capture_output=True, check=True).stdout.decode().strip().split("\n")
    assert out[0] == "B 1"                      # the one-slot hand-over contract (SetPointHandOver)
    rows = [l.split() for l in out[1:]]
    assert len(rows) == len(pts)
    c = oracle.OracleConsumer()
    c.reset(start)
    for k, p in enumerate(pts):
        if k == double_fill_at:
            assert c.fill(p) and not c.fill(p)   # the second put() before a get() fails, trajectory_buffer.cpp:45-48
            cycles = 0
            while True:
                c.update(velocity / 0.9)
                cycles += 1
                if c.ready():
                    break
        else:
            cycles = c.deliver(p, velocity, 200000)
        st, cn = c.state()
        r = rows[k]
        assert int(r[0]) == k and int(r[1]) == cycles, (k, r[1], cycles)
        assert [int(x) for x in r[2:8]] == cn, (k, r[2:8], cn)
        np.testing.assert_array_equal(np.array(r[8:23], dtype=float), st)
    return c.state()


@pytest.mark.parametrize("task", ["dual_arms_static1", "dual_arms_dyn1"])
def test_planner_set_points_through_the_consumer(consumer_exe, oracle, scenes, tmp_path, task):
    sc, pts = planner_set_points(oracle, scenes, task, 400)
    # taskCallback publishes the initial position + 1e-5 in z first (panda_bimanual_control.cpp:514-518)
    first = sc["start"] + np.array([0.0, 0.0, 0.00001])
    seq = np.vstack([first[None], pts])
    st, cn = run_both(consumer_exe, oracle, tmp_path, seq, sc["start"], sc["velocity_max"])
    accepted, refused, n_nan, too_close, inconsistent, updates = cn
    # what the planner emits satisfies the contract: every point taken, none refused / NaN / too close
    assert accepted == len(seq) and refused == 0 and n_nan == 0 and too_close == 0
    # v_goal = min(|d| * 100, 0.9 * v_max) with v_max = velocity / 0.9 (vrep_controller.cpp:291-292)
    assert 0 < st[0] <= sc["velocity_max"] * (1 + 1e-12)
    print("%s: %d set-points, %d controller cycles, %d 'inconsistent trajectory' warnings" % (task, len(seq), updates, inconsistent))


def test_consumer_edge_cases(consumer_exe, oracle, tmp_path):
    start = np.array([0.1, -0.2, 0.5])
    pts = [start + [0.002, 0, 0], start + [0.004, 0.001, 0],
           start + [0.004, 0.001, 0],                 # identical: nudged by +2e-6 in z
           start + [0.004, 0.001, 2e-6 + 3e-7],       # < 1e-6 from the nudged point: nudged by -2e-6
           start + [0.006, 0.0, 0.001],
           [np.nan, 0.0, 0.0],                        # "Planner sent NaN."
           start + [0.008, 0.0, 0.001]]               # never taken: the NaN poisoned next_ng (as in the reference)
    st, cn = run_both(consumer_exe, oracle, tmp_path, np.asarray(pts[:5]), start, 0.2)
    assert cn[3] == 2 and cn[2] == 0
    st, cn = run_both(consumer_exe, oracle, tmp_path, np.asarray(pts[:6]), start, 0.2)
    assert cn[2] == 1
    # a planner that sends two points without waiting: the second one is refused
    st, cn = run_both(consumer_exe, oracle, tmp_path, np.asarray(pts[:2] + pts[4:5]), start, 0.2, double_fill_at=1)
    assert cn[1] == 1
    # a jump back behind the nominal goal makes the radicand negative: "Inconsistent trajectory detected."
    # a fast segment (nominal goal overshoots its end by up to v * 1 ms) followed by a tiny perpendicular one
    # (v_goal = |d| * 100 is small): the radicand goes negative -> "Inconsistent trajectory detected.", b = 0
    zig = [start + [0.0503, 0, 0], start + [0.0503, 1e-5, 0.0], start + [0.08, 0.01, 0.0]]
    st, cn = run_both(consumer_exe, oracle, tmp_path, np.asarray(zig), start, 0.2)
    assert cn[4] >= 1
    print("edge cases: counters", cn)

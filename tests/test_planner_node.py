"""Rows f1/f2/f4 of SURVEY.md 8(f): task-file loader, planner-node mirror,
obstacle stream and set-point validator (include/bimanual_planning_ros/
planner_node.h) through the head-less driver tools/plan_task."""
import glob
import json
import os
import subprocess

import numpy as np
import pytest

import conftest

ROOT = conftest.ROOT
EXE = conftest.exe(os.path.join(ROOT, "tools", "plan_task"))
TASKS = os.path.join(ROOT, "tests", "golden", "tasks")


def _dump(path):
    out = subprocess.run([EXE, path, "--dump-params"], capture_output=True, check=True).stdout
    return json.loads(out)


def _check_against_yaml(path):
    yaml = pytest.importorskip("yaml")
    ref = yaml.safe_load(open(path))["bimanual_planning"]
    got = _dump(path)
    for k in ("num_agents_ee", "num_agents_body", "k_attr", "k_circ", "k_repel", "k_damp", "k_manip", "k_repel_body",
              "k_goal_dist", "k_path_len", "k_safe_dist", "k_workspace", "max_prediction_steps", "approach_dist",
              "detect_shell_rad", "prediction_freq_multiple", "frequency_ros", "velocity", "open_loop"):
        assert got[k] == ref[k], (path, k)
    assert got["desired_ws_limits"] == [float(x) for x in ref["desired_ws_limits"]]
    assert len(got["obstacles"]) == len(ref["obstacles"])
    for g, r in zip(got["obstacles"], ref["obstacles"]):
        assert g == [float(x) for x in r["pos"]] + [float(x) for x in r.get("vel", [0, 0, 0])] + [float(r["radius"])]
    assert len(got["goals"]) == len(ref["goals"])
    for g, r in zip(got["goals"], ref["goals"]):
        assert g["type"] == r["type"] and g["end_condition"] == r.get("end_condition", "")
        if "pos" in r:
            assert g["pos"] == [float(x) for x in r["pos"]]
        for k, v in g["overrides"].items():
            assert float(r[k]) == v


def test_task_loader_on_fixtures(hip_lib):
    for f in sorted(glob.glob(os.path.join(TASKS, "*.yaml"))):
        _check_against_yaml(f)
    d = _dump(os.path.join(TASKS, "dyn1.yaml"))
    assert d["k_circ"] == 0.025 and d["goals"][1]["overrides"] == {"k_circ": 0.015}


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/bimanual_planning_ros/config/tasks"),
                    reason="reference checkout only exists in the build container")
def test_task_loader_on_the_reference_task_files(hip_lib):
    files = sorted(glob.glob("/root/reference/src/bimanual_planning_ros/config/tasks/*.yaml"))
    assert len(files) == 9
    for f in files:
        _check_against_yaml(f)


def _oracle_node_run(oracle, scenes, sc, max_ticks):
    """the same loop as tools/plan_task.cpp, on the oracle"""
    ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])  # init() after the initial position was recorded
    ora.set_initial_position(sc["start"])
    obs = sc["obstacles"].copy()
    rows = []
    for t in range(max_ticks):
        b = ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        rows.append((t, b) + tuple(ora.real_state()[0]) + (ora.dist_from_goal(),))
        obs = scenes.advance_live_obstacles(obs)
        if ora.dist_from_goal() < 0.01:
            break
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["static1", "dyn1"])
def test_headless_task_run_matches_oracle(pmaf, oracle, scenes, tmp_path, hip_lib, task):
    oracle.set_exp_mode(1)
    try:
        sc = scenes.static1_scene(10, 300) if task == "static1" else scenes.dyn1_scene(10, 600)
        rvf = tmp_path / "rv.bin"
        np.ascontiguousarray(sc["random_vecs"]).tofile(rvf)
        cmd = [EXE, os.path.join(TASKS, task + ".yaml"), "--start"] + [repr(float(x)) for x in sc["start"]] + \
              ["--max-ticks", "1200", "--random-vecs", str(rvf), "--consumer"]
        r = subprocess.run(cmd, capture_output=True, check=True, env=conftest.binary_env(pmaf))
        lines = [l for l in r.stdout.decode().strip().split("\n")]
        rows = _oracle_node_run(oracle, scenes, sc, 1200)
        data = [l for l in lines if not l.startswith("#") and not l.startswith("C ")]
        cons = [l.split() for l in lines if l.startswith("C ")]
        assert len(data) == len(rows) == len(cons)
        # f4: every set-point through the oracle's restatement of the consumer (TrajectoryBuffer + followTrajectory,
        # costp_controller.cpp:289-344): same controller cycles, v_goal, next_ng and counters per tick
        oc = oracle.OracleConsumer()
        oc.reset(sc["start"])
        oc.deliver(sc["start"] + np.array([0.0, 0.0, 0.00001]), sc["velocity_max"])   # :514-518
        for c, ro in zip(cons, rows):
            cycles = oc.deliver(np.array(ro[2:5]), sc["velocity_max"])
            st, cn = oc.state()
            assert int(c[1]) == ro[0] and int(c[2]) == cycles
            assert float(c[3]) == st[0] and float(c[4]) == st[2]
            assert [int(x) for x in c[5:10]] == cn[:5]
        assert cn[1] == 0 and cn[2] == 0 and cn[3] == 0   # nothing refused, no NaN, no point closer than 1e-6 m
        for l, ro in zip(data, rows):
            f = l.split()
            assert int(f[0]) == ro[0] and int(f[1]) == ro[1]
            assert [float(x) for x in f[2:6]] == list(ro[2:6])
        assert lines[-1].startswith("# goal reached")
        assert b"rejected" not in r.stderr  # every set-point satisfied the consumer contract (f4)
    finally:
        oracle.set_exp_mode(0)

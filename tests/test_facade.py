"""The C++ facade ghostplanner::cfplanner::CfManager / Obstacle
(include/bimanual_planning_ros/) driven like the reference's planner node
(tests/cpp/facade_tick.cpp) must reproduce the oracle."""
import os
import subprocess

import numpy as np
import pytest

import conftest

ROOT = conftest.ROOT


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
        exe = os.path.join(ROOT, "tests", "cpp", "facade_tick")
        out = subprocess.run([exe, str(N), str(cap), str(ticks), str(rvf)], capture_output=True, check=True).stdout.decode()
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

#!/usr/bin/env python3
"""Writes oracle/pin/scenarios/*.txt: the inputs of the reference-pin runs (oracle/pin/pin_harness.cpp), from the same
scene definitions the parity tests use (predictive-multi-agent-framework_amd/scenes.py, tests/golden/task_scenes.json).
Every double is written as a C99 hex literal. The Random agents' vectors are handed over RAW (U(-1,1)^3 triples in the
order the reference's constructors call makeRandomVector(): agents 5 .. N-1, one triple per obstacle) -- the reference
normalises them itself (B/include/bimanual_planning_ros/cf_agent.h:338-342), and tests/test_reference_pin.py
normalises the same triples in the association it is testing.

TEST INFRASTRUCTURE. usage: python oracle/pin/make_scenarios.py [--out DIR]
(default DIR: oracle/pin/scenarios, the committed files; tests/test_reference_pin.py writes into a temporary directory and
compares -- a test run never rewrites tracked files)"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

OUT = os.path.join(HERE, "scenarios")
h = lambda x: float(x).hex()


def raw_triples(S, seed, n_agents, n_obs, n_inits):
    """the SplitMix64 stream scenes.random_unit_vectors() draws from, un-normalised: block k = init() number k"""
    rng = S.SplitMix64(seed)
    return [rng.uniform_range(-1.0, 1.0, n_agents * n_obs * 3).reshape(n_agents, n_obs, 3) for _ in range(n_inits)]


def emit(S, name, sc, goals, seed, dynamic=False, lag=None, mult=1, dump_paths=0, detail_every=1):
    """goals: [(xyz, max_ticks, until_reached)]"""
    N, n_obs = int(sc["n_agents"]), sc["obstacles"].shape[0]
    L = ["pmaf-pin-scenario 1", "name " + name, "n_agents %d" % N, "n_body 1",
         "max_prediction_steps %d" % int(sc["max_prediction_steps"]), "freq_multiple %d" % mult, "dt " + h(sc["dt"]),
         "gains " + " ".join(h(sc[k]) for k in ("k_attr", "k_circ", "k_repel", "k_damp")) + " " + h(0.0),
         "k_repel_body " + h(0.02),
         "limits " + " ".join(h(sc[k]) for k in ("velocity_max", "approach_dist", "detect_shell_rad")),
         "cost " + " ".join(h(x) for x in sc["cost_gains"]), "ws " + " ".join(h(x) for x in sc["ws_limits"]),
         "start " + " ".join(h(x) for x in sc["start"]),
         "closed_loop %d %s" % (1 if lag is not None else 0, h(lag or 0.0)), "dynamic %d" % (1 if dynamic else 0),
         "dump_paths %d" % dump_paths, "detail_every %d" % detail_every, "obstacles %d" % n_obs]
    L += [" ".join(h(x) for x in row) for row in sc["obstacles"]]
    L.append("goals %d" % len(goals))
    L += ["%s %d %d" % (" ".join(h(x) for x in g), ticks, 1 if reached else 0) for g, ticks, reached in goals]
    blocks = raw_triples(S, seed, N, n_obs, len(goals))
    rnd = np.concatenate([b[5:].reshape(-1, 3) for b in blocks]) if N > 5 else np.zeros((0, 3))
    L.append("random %d" % len(rnd))
    L += [" ".join(h(x) for x in r) for r in rnd]
    open(os.path.join(OUT, name + ".txt"), "w").write("\n".join(L) + "\n")


if __name__ == "__main__":
    if "--out" in sys.argv:
        OUT = os.path.abspath(sys.argv[sys.argv.index("--out") + 1])
    S = graft.load_package().scenes
    os.makedirs(OUT, exist_ok=True)
    tasks = json.load(open(os.path.join(ROOT, "tests", "golden", "task_scenes.json")))
    s1 = 0xC0FFEE00 + 1 * 256
    s6 = 0xC0FFEE00 + 6 * 256
    s9 = 0xC0FFEE00 + 9 * 256
    # the reference's own CPU-runnable case and BASELINE C1 (static1, N = 10 as shipped / N = 16)
    sc = S.static1_scene(10, 100)
    emit(S, "static1_n10_h100", sc, [(sc["goal"], 31, False)], s1, dump_paths=1, detail_every=6)   # full paths on 6 ticks
    sc = S.static1_scene(16, 100)
    emit(S, "c1_static1_n16_h100", sc, [(sc["goal"], 25, False)], s1, dump_paths=1, detail_every=6)
    # the task file as shipped: max_prediction_steps 1500, until `reached` (early stops at distGoal <= 0.1)
    sc = S.static1_scene(10, 1499)
    emit(S, "static1_shipped", sc, [(sc["goal"], 2000, True)], s1, detail_every=25)
    # moving obstacles (predictObstacles + the obstacle stream), until `reached`: SURVEY's second probe
    sc = S.dyn1_scene(10, 1499)
    emit(S, "dyn1_shipped", sc, [(sc["goal"], 2000, True)], s6, dynamic=True, detail_every=25)
    # 22 obstacles
    sc = S.scene_from_record(tasks["dual_arms_trap"], "trap")
    emit(S, "trap_shipped", sc, [(sc["goal"], 900, True)], s9, detail_every=25)
    # a hysteresis case: dyn1 at a 300-step horizon switches 0 -> 3 -> 8 -> 2 (a Random agent leads for 30 ticks)
    sc = S.dyn1_scene(10, 300)
    emit(S, "dyn1_h300_hysteresis", sc, [(sc["goal"], 2000, True)], s6, dynamic=True, detail_every=25)
    # BASELINE C2, the configuration the headline number is quoted on
    sc = S.config_scene("C2")
    emit(S, "c2_64x200x32", sc, [(sc["goal"], 12, False)], 0xC0FFEE00 + 2 * 256, detail_every=1)
    # the node's other boundary paths (tests/test_boundary_gpu.py)
    sc = S.static1_scene(10, 300)
    emit(S, "static1_closed_loop_lag30", sc, [(sc["goal"], 2500, True)], s1, lag=0.3, detail_every=25)
    sc = S.dyn1_scene(10, 300)
    emit(S, "dyn1_two_goals", sc, [(sc["goal"], 2000, True), (np.array([-0.45, 0.1, 0.6]), 2000, True)], s6, dynamic=True,
         detail_every=25)
    emit(S, "dyn1_freq2", sc, [(sc["goal"], 400, False)], s6, dynamic=True, mult=2, detail_every=25)
    print("wrote", sorted(os.listdir(OUT)))

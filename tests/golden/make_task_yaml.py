#!/usr/bin/env python3
"""Writes the task-file fixtures tests/golden/tasks/*.yaml in the schema of the
reference's task files (root key `bimanual_planning`)
from the scene definitions in predictive-multi-agent-framework_amd/scenes.py.
Only the keys the planner node reads are emitted."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def emit(sc, path, n_agents, max_steps, extra_goal_keys=""):
    o = sc["obstacles"]
    L = ["bimanual_planning:",
         "    num_agents_ee: %d" % n_agents, "    num_agents_body: 1",
         "    k_attr: %r" % float(sc["k_attr"]), "    k_circ: %r" % float(sc["k_circ"]),
         "    k_repel: %r" % float(sc["k_repel"]), "    k_damp: %r" % float(sc["k_damp"]),
         "    k_manip: 0.0", "    k_repel_body: 0.02",
         "    k_goal_dist: %r" % float(sc["cost_gains"][0]), "    k_path_len: %r" % float(sc["cost_gains"][1]),
         "    k_safe_dist: %r" % float(sc["cost_gains"][2]), "    k_workspace: %r" % float(sc["cost_gains"][3]),
         "    desired_ws_limits: [%s]" % ", ".join(repr(float(x)) for x in sc["ws_limits"]),
         "    max_prediction_steps: %d" % max_steps,
         "    approach_dist: %r" % sc["approach_dist"], "    detect_shell_rad: %r" % sc["detect_shell_rad"],
         "    prediction_freq_multiple: 1", "    frequency_ros: 100", "    velocity: %r" % sc["velocity_max"],
         "    open_loop: true", "    visualize_commanded_path: true", "    visualize_predicted_paths: true",
         "    obstacles:"]
    for i, r in enumerate(o):
        if i == len(o) - 1:
            L.append("      # Repulsive obstacle for self collision avoidance")
        L += ["      - pos: [%r, %r, %r]" % tuple(float(x) for x in r[0:3]),
              "        radius: %r" % float(r[6]),
              "        vel: [%r, %r, %r]" % tuple(float(x) for x in r[3:6])]
    L += ["    goals:", "      - type: key", "        message: \"Press start planning.\"",
          "      - type: plan", "        pos: [%r, %r, %r]" % tuple(float(x) for x in sc["goal"]),
          "        end_condition: reached"]
    if extra_goal_keys:
        L.append(extra_goal_keys)
    open(path, "w").write("\n".join(L) + "\n")


if __name__ == "__main__":
    pkg = graft.load_package()
    out = os.path.join(ROOT, "tests", "golden", "tasks")
    emit(pkg.scenes.static1_scene(10, 300), os.path.join(out, "static1.yaml"), 10, 301)
    d = pkg.scenes.dyn1_scene(10, 600)
    d["k_circ"] = 0.025  # the goal overrides it back to dyn1's 0.015 (per-goal override path)
    emit(d, os.path.join(out, "dyn1.yaml"), 10, 601, "        k_circ: 0.015")
    print("wrote", os.listdir(out))

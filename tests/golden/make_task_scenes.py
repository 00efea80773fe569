#!/usr/bin/env python3
"""tests/golden/task_scenes.json: the planner inputs of the nine task files the
reference ships (config/tasks/*.yaml) as plain data -- gains, limits, obstacle
lists, goal of the `plan` goal -- plus a start position chosen here (the YAML
has none: it comes from the robot's forward kinematics). Run in the build
container only (needs /root/reference); the JSON travels."""
import glob
import json
import os

import numpy as np
import yaml

REF = "/root/reference/src/bimanual_planning_ros/config/tasks"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "task_scenes.json")

scenes = {}
for f in sorted(glob.glob(os.path.join(REF, "*.yaml"))):
    d = yaml.safe_load(open(f))["bimanual_planning"]
    goal = next(g for g in d["goals"] if g["type"] == "plan")
    obs = [[*map(float, o["pos"]), *map(float, o.get("vel", [0, 0, 0])), float(o["radius"])] for o in d["obstacles"]]
    g = np.array(goal["pos"], dtype=float)
    o0 = np.array(obs[0][:3])
    # start: 0.6 m beyond the first obstacle as seen from the goal, 4 cm to the side, clamped into the workspace box
    u = (o0 - g) / np.linalg.norm(o0 - g)
    ws = [float(x) for x in d["desired_ws_limits"]]
    side = np.cross(u, [0.0, 0.0, 1.0])
    side = side / np.linalg.norm(side)
    start = o0 + 0.6 * u + 0.04 * side  # off the obstacle-goal line (the Had heuristic is singular on it)
    start = np.minimum(np.maximum(start, [ws[1] + 0.05, ws[3] + 0.02, ws[5] + 0.05]), [ws[0] - 0.05, ws[2] - 0.02, ws[4] - 0.05])
    scenes[os.path.basename(f)[:-5]] = {
        "n_agents": int(d["num_agents_ee"]), "max_prediction_steps": int(d["max_prediction_steps"]),
        "dt": float(d["prediction_freq_multiple"]) / float(d["frequency_ros"]),
        "velocity_max": float(d["velocity"]), "approach_dist": float(d["approach_dist"]),
        "detect_shell_rad": float(d["detect_shell_rad"]),
        "k_attr": float(d["k_attr"]), "k_circ": float(d["k_circ"]), "k_repel": float(d["k_repel"]),
        "k_damp": float(d["k_damp"]),
        "cost_gains": [float(d["k_goal_dist"]), float(d["k_path_len"]), float(d["k_safe_dist"]), float(d["k_workspace"])],
        "ws_limits": ws, "obstacles": obs, "goal": [float(x) for x in g],
        "start": [round(float(x), 3) for x in start],
    }
json.dump(scenes, open(OUT, "w"), indent=1)
print("wrote", OUT, list(scenes))

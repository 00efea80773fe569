"""Per-agent rollout durations (pmaf_get_prediction_times_ns) by heuristic type:
shows which agent's wave bounds the tick in the wave-per-agent kernel.
usage: python tools/agenttime.py [C2 C3 ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
NAMES = {1: "GOAL", 2: "OBST", 3: "GOALOBST", 4: "VEL", 5: "RANDOM", 6: "HAD"}
for name in (sys.argv[1:] or ["C2"]):
    sc = pm.scenes.config_scene(name)
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    h.set_initial_position(sc["start"])
    types = pm.scenes.default_agent_types(sc["n_agents"])
    acc = []
    for k in range(40):
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop()
        if k >= 8:
            acc.append(np.asarray(h.prediction_times_ns()).reshape(-1))
    t = np.mean(acc, axis=0) / 1e3
    n = h.n_points()[0] if hasattr(h, "n_points") else None
    print(name, "per-agent rollout us: max %.1f  median %.1f  min %.1f" % (t.max(), np.median(t), t.min()))
    for ty in sorted(set(types.tolist())):
        sel = t[types == ty]
        print("  %-9s n=%3d  mean %.1f  max %.1f" % (NAMES.get(ty, ty), len(sel), sel.mean(), sel.max()))
    order = np.argsort(-t)[:8]
    print("  slowest:", [(int(i), NAMES.get(int(types[i])), round(float(t[i]), 1)) for i in order])
    h.close()

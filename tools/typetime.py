"""per-heuristic cost of the rollout: populations of ONE agent type each (same C2 scene, same Random vectors),
per-agent rollout durations from the device clock"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
NAMES = {1: "GOAL", 2: "OBST", 3: "GOALOBST", 4: "VEL", 5: "RANDOM", 6: "HAD"}
for ty in (5, 1, 2, 3, 4, 6):
    sc = pm.scenes.config_scene("C2")
    sc["agent_types"] = np.full(sc["n_agents"], ty, dtype=np.int32)
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"]); h.set_profiling(True)
    acc = []
    for k in range(40):
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"]); h.stop()
        if k >= 8: acc.append(np.asarray(h.prediction_times_ns()).reshape(-1))
    t = np.mean(acc, axis=0) / 1e3
    ms, n, steps = h.kernel_stats()
    print("%-9s per-agent us: mean %.1f max %.1f min %.1f | kernel %.1f us | distinct paths %d" % (NAMES[ty], t.mean(), t.max(), t.min(), ms / n * 1e3, len(set(np.round(t, 1)))), flush=True)
    h.close()

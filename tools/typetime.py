import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
NAMES = {1: "GOAL", 2: "OBST", 3: "GOALOBST", 4: "VEL", 5: "RANDOM", 6: "HAD"}
for ty in (5, 1, 3, 2, 4, 6):
    sc = pm.scenes.synthetic_scene(64, 200, 32, 2, 0, agent_types=np.full(64, ty, dtype=np.int32))
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"]); h.set_profiling(True)
    for _ in range(10): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); h.reset_kernel_stats()
    for k in range(60):
        if k % 20 == 0: h.set_initial_position(sc["start"])
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); ms, n, steps = h.kernel_stats()
    t = np.asarray(h.prediction_times_ns()).reshape(-1) / 1e3
    print(NAMES[ty], "kernel %.1f us" % (ms / n * 1e3), "agent-steps per launch %.0f" % (steps / n), "last rollout per-agent us: max %.1f mean %.1f" % (t.max(), t.mean()), flush=True)
    h.close()

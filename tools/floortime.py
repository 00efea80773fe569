"""kernel-time floor: the C2 population with a 1-step horizon (launch + prologue + epilogue only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
for cap in (2, 11, 201):
    sc = dict(pm.scenes.config_scene("C2")); sc["max_prediction_steps"] = cap
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"]); h.set_profiling(True)
    for _ in range(20): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); h.reset_kernel_stats(); t0 = time.perf_counter(); K = 200
    for _ in range(K): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); t1 = time.perf_counter(); ms, n, steps = h.kernel_stats()
    print("cap", cap, "tick %.1f us kernel %.1f us" % ((t1 - t0) / K * 1e6, ms / n * 1e3), flush=True)
    h.close()

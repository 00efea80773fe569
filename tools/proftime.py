"""tick time with / without the HIP-event timing of the rollout launches (pmaf_set_profiling)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
sc = pm.scenes.config_scene(sys.argv[1] if len(sys.argv) > 1 else "C2")
for rep in range(2):
    for prof in (False, True):
        h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"])
        h.set_profiling(prof)
        for _ in range(30): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop(); t0 = time.perf_counter(); K = 400
        for _ in range(K): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop(); t1 = time.perf_counter()
        print("profiling %s: tick %.2f us" % (prof, (t1 - t0) / K * 1e6), flush=True)
        h.close()

"""tick rate with / without the HIP profiling events around the rollout kernel,
and the set-point latency of a tick issued on an idle stream (previous rollout
finished): host call -> best index + next set-point on the host."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
for name in (sys.argv[1:] or ["C2"]):
    sc = pm.scenes.config_scene(name)
    for prof in (False, True):
        h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"])
        h.set_profiling(prof)
        for _ in range(50): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop(); K = 500; t0 = time.perf_counter()
        for _ in range(K): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop(); t1 = time.perf_counter()
        msg = "%s profiling=%d tick %.1f us" % (name, prof, (t1 - t0) / K * 1e6)
        if prof:
            ms, n, steps = h.kernel_stats(); msg += " kernel %.1f us" % (ms / n * 1e3)
        print(msg, flush=True)
        if not prof:
            lat = []
            for _ in range(200):
                h.stop()
                ta = time.perf_counter(); h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"]); lat.append(time.perf_counter() - ta)
            lat = np.array(lat) * 1e6
            print("%s idle-start set-point latency: median %.1f us  p99 %.1f us  min %.1f us" % (name, np.median(lat), np.percentile(lat, 99), lat.min()), flush=True)
        h.close()

"""mean over ticks of the slowest agent's rollout duration vs the kernel duration (HIP events)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
sc = pm.scenes.config_scene(sys.argv[1] if len(sys.argv) > 1 else "C2")
h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"]); h.set_profiling(True)
mx, arg = [], []
for k in range(220):
    h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"]); h.stop()
    if k == 19: h.reset_kernel_stats()
    if k >= 20:
        t = np.asarray(h.prediction_times_ns()).reshape(-1); mx.append(t.max() / 1e3); arg.append(int(t.argmax()))
ms, n, _ = h.kernel_stats()
print("mean of per-tick max agent %.1f us; kernel (events) %.1f us; slowest agent histogram" % (np.mean(mx), ms / n * 1e3), np.bincount(arg).argsort()[::-1][:6], np.sort(np.bincount(arg))[::-1][:6])
h.close()

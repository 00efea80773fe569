"""Launch duration (HIP events on the kernel's dispatch packet) against the slowest wave's own duration (device clock),
over agent count / horizon / obstacle count: what a wave-per-agent launch costs beyond its slowest wave.
usage: python tools/launchgap.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
for (N, H, M) in [(256, 500, 128), (256, 250, 128), (128, 500, 128), (64, 500, 128), (16, 500, 128), (256, 500, 61), (256, 100, 128), (1024, 100, 32), (64, 200, 32), (16, 200, 32)]:
    sc = pm.scenes.synthetic_scene(N, H, M, 3, 0)
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"], lanes_per_agent=64)
    h.set_initial_position(sc["start"])
    for k in range(6):
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); h.set_profiling(True); h.reset_kernel_stats()
    acc = []
    for k in range(20):
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"]); h.stop()
        acc.append(np.asarray(h.prediction_times_ns()).reshape(-1).max() / 1e3)
    ms, n, steps = h.kernel_stats()
    ker = ms / n * 1e3
    print("N %4d H %3d M %3d  h_eff %.0f  kernel %.1f us  slowest wave %.1f us  gap %.1f us  path bytes %.0f KB" % (
        N, H, M, steps / n / N, ker, np.mean(acc), ker - np.mean(acc), N * (H + 1) * 24 / 1024))
    h.close()

"""rollout-kernel time of P populations of a BASELINE config in one handle for each lanes-per-agent mapping
(calibrates pick_lpa). usage: python tools/lpasweep.py C2 16,32 64,32,16"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
cfg = sys.argv[1]
Ps = [int(x) for x in sys.argv[2].split(",")]
lpas = [int(x) for x in sys.argv[3].split(",")]
for P in Ps:
    scs = [pm.scenes.config_scene(cfg, scene_id=i) for i in range(P)]
    starts = np.stack([s["start"] for s in scs]); sc = scs[0]
    for lpa in lpas:
        h = pm.PmafPlanner(scs, device=0, mgr_init_pos=starts, lanes_per_agent=lpa); h.set_initial_position(starts); h.set_profiling(True)
        for _ in range(5): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop(); h.reset_kernel_stats(); K = 30; t0 = time.perf_counter()
        for _ in range(K): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop(); t1 = time.perf_counter(); ms, n, st = h.kernel_stats()
        print(cfg, "P", P, "agents", P * sc["n_agents"], "lpa", lpa, h.launch_config(), "kernel %.1f us rollouts/s %.0f" % (ms / n * 1e3, P * sc["n_agents"] * K / (t1 - t0)), flush=True)
        h.close()

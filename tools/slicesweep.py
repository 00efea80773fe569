"""The wave-per-agent kernel with two waves on a SIMD: kernel us per launch without / with the priority-slicing loop
(PMAF_W64_SLICE, csrc/pmaf_k_w64.hip SLICE) over slice lengths and shares. Interleaved repeats, median.
usage: python tools/slicesweep.py M:N:P [...]      (H, REPS, SETTINGS="log2:younger,..." from the environment)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
H = int(os.environ.get("H", "200"))
REPS = int(os.environ.get("REPS", "3"))
SETTINGS = [tuple(int(x) for x in s.split(":")) for s in os.environ.get("SETTINGS", "9:6,9:5,9:4,8:5,10:5,10:6,8:6").split(",")]
for a in sys.argv[1:]:
    M, N, P = (int(x) for x in a.split(":"))
    if M == 0:   # BASELINE C5's scenes
        scs = [pm.scenes.config_scene("C5", scene_id=i) for i in range(P)]
        N, M = scs[0]["n_agents"], scs[0]["obstacles"].shape[0] - 1
    else:
        scs = [pm.scenes.synthetic_scene(N, H, M, 3, i) for i in range(P)]
    sc = scs[0]; starts = np.stack([s["start"] for s in scs])
    arg, st = (scs, starts) if P > 1 else (sc, sc["start"])
    cases = [("off", None)] + [("%d:%d" % s, s) for s in SETTINGS]
    us = {c[0]: [] for c in cases}
    for rep in range(REPS):
        for name, s in cases:
            os.environ["PMAF_W64_SLICE"] = "0" if s is None else "1"
            if s is not None:
                os.environ["PMAF_W64_SLICE_LOG2"], os.environ["PMAF_W64_SLICE_YOUNGER"] = str(s[0]), str(s[1])
            h = pm.PmafPlanner(arg, device=0, mgr_init_pos=st, lanes_per_agent=64)
            h.set_initial_position(st); h.set_profiling(True)
            for _ in range(4): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
            h.stop(); h.reset_kernel_stats()
            for _ in range(12): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
            h.stop(); ms, n, steps = h.kernel_stats()
            us[name].append(ms / n * 1e3)
            h.close()
    print("M %3d N %5d P %d H %d | %s" % (M, N, P, sc["max_prediction_steps"] - 1, " | ".join("%s %.0f" % (c[0], float(np.median(us[c[0]]))) for c in cases)), flush=True)

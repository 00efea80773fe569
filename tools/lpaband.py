"""kernel us per launch over lanes-per-agent mappings for P populations x N agents x M field obstacles in ONE handle
(round 6: the 1 025 ... 2 048-wave band of pick_lpa, i.e. between one and two waves per SIMD of the wave-per-agent kernel).
usage: python tools/lpaband.py M:N:P:lpa,lpa,... [...]   (lpa 0 = the library's choice; H from the environment, default 200)
Interleaved repeats (REPS, default 3), the median per mapping is printed: boxes differ by +- 2 %, runs on one box by +- 0.3 %."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
H = int(os.environ.get("H", "200"))
REPS = int(os.environ.get("REPS", "3"))
for a in sys.argv[1:]:
    M, N, P, l = a.split(":")
    M, N, P = int(M), int(N), int(P)
    scs = [pm.scenes.synthetic_scene(N, H, M, 3, i) for i in range(P)]
    sc = scs[0]; starts = np.stack([s["start"] for s in scs])
    arg = scs if P > 1 else sc
    lpas = [int(x) for x in l.split(",")]
    us = {lpa: [] for lpa in lpas}; name = {}
    for rep in range(REPS):
        for lpa in lpas:
            try:
                h = pm.PmafPlanner(arg, device=0, mgr_init_pos=starts if P > 1 else sc["start"], lanes_per_agent=lpa)
            except pm.PmafError:
                name[lpa] = "lpa %d refused" % lpa; continue
            h.set_initial_position(starts if P > 1 else sc["start"]); h.set_profiling(True)
            for _ in range(4): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
            h.stop(); h.reset_kernel_stats()
            for _ in range(12): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
            h.stop(); ms, n, steps = h.kernel_stats(); cfg = h.launch_config()
            us[lpa].append(ms / n * 1e3)
            name[lpa] = "lpa %d%s %s" % (cfg["lanes_per_agent"], " (auto)" if lpa == 0 else "", cfg.get("kernel", ""))
            h.close()
    row = ["%s: %.0f us" % (name[lpa], float(np.median(us[lpa]))) if us[lpa] else name[lpa] for lpa in lpas]
    print("M %3d N %5d P %d H %d | %s" % (M, N, P, H, " | ".join(row)), flush=True)

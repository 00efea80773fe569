"""M > 128 routing: the four-slot wave-per-agent kernel (k_rollout_w64<4>: 256 VGPR + AGPR spill space, occupancy 1)
against the generic LDS-table kernel and the lane-group mappings, same scene (M = 200 / 256 obstacles).
usage: python tools/m200time.py [N] [H]"""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import __graft_entry__ as g
    pm = g.load_package()
    N, H, M, lpa = (int(x) for x in sys.argv[2:6])
    sc = pm.scenes.synthetic_scene(N, H, M, 3, 7)
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"], lanes_per_agent=lpa)
    h.set_initial_position(sc["start"]); h.set_profiling(True)
    for _ in range(5): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); h.reset_kernel_stats()
    for _ in range(30): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); ms, n, steps = h.kernel_stats()
    print("N %d H %d M %d lpa %d generic=%s: kernel %.1f us" % (N, H, M, lpa, os.environ.get("PMAF_FORCE_GENERIC", "0"), ms / n * 1e3), flush=True)
    sys.exit(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 300
CASES = ((64, "0"), (64, "1"), (32, "1"), (16, "1")) if os.environ.get("M200_ALL", "1") == "1" else ((0, "0"), (64, "0"))
for M in (200, 256):
    for lpa, gen in CASES:
        env = dict(os.environ, PMAF_FORCE_GENERIC=gen)
        subprocess.run([sys.executable, __file__, "--child", str(N), str(H), str(M), str(lpa)], env=env)

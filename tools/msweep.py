"""kernel time over the number of field obstacles M (64 agents x 200 steps, wave-per-agent kernel)
usage: PMAF_LIB_PATH=... python tools/msweep.py 4 9 16 24 32 48 61"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
for M in [int(x) for x in sys.argv[1:]] or [4, 9, 16, 24, 32, 48, 61]:
    sc = pm.scenes.synthetic_scene(64, 200, M, 2, 1)
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"]); h.set_profiling(True)
    for _ in range(20): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); h.reset_kernel_stats()
    for _ in range(200): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); ms, n, steps = h.kernel_stats()
    print("M %3d kernel %.1f us" % (M, ms / n * 1e3), flush=True)
    h.close()

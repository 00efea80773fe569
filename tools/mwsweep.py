"""timing experiments: kernel us per launch over obstacle counts / agent counts for the PMAF_MW setting in the environment
usage: PMAF_MW=0|1|3|4 python tools/mwsweep.py M[:N] ..."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
for a in sys.argv[1:]:
    M, N = (int(x) for x in (a.split(":") + ["256"])[:2])
    sc = pm.scenes.synthetic_scene(N, 300, M, 3, 0)
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"]); h.set_profiling(True)
    for _ in range(10): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); h.reset_kernel_stats()
    for _ in range(60): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); ms, n, steps = h.kernel_stats()
    print("M", M, "N", N, "PMAF_MW", os.environ.get("PMAF_MW"), "kernel %.1f us" % (ms / n * 1e3), "us/step %.3f" % (ms / n * 1e3 / 299), flush=True)
    h.close()

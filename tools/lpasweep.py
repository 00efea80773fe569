"""kernel us per launch over lanes-per-agent mappings for synthetic populations (many agents x many obstacles):
which mapping pick_lpa should choose. usage: python tools/lpasweep.py M:N:lpa,lpa,... [...]   (lpa 0 = the library's choice)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
H = int(os.environ.get("H", "200"))
for a in sys.argv[1:]:
    M, N, l = a.split(":")
    M, N = int(M), int(N)
    sc = pm.scenes.synthetic_scene(N, H, M, 3, 0)
    row = []
    for lpa in (int(x) for x in l.split(",")):
        try:
            h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"], lanes_per_agent=lpa)
        except pm.PmafError as e:
            row.append("lpa %d refused" % lpa); continue
        h.set_initial_position(sc["start"]); h.set_profiling(True)
        for _ in range(4): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop(); h.reset_kernel_stats()
        for _ in range(12): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop(); ms, n, steps = h.kernel_stats(); cfg = h.launch_config()
        row.append("lpa %d%s: %.0f us" % (cfg["lanes_per_agent"], " (auto)" if lpa == 0 else "", ms / n * 1e3))
        h.close()
    print("M %3d N %5d H %d | %s" % (M, N, H, " | ".join(row)), flush=True)

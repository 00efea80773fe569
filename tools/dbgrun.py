import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = pm.scenes.config_scene(name)
h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
h.set_initial_position(sc["start"])
for k in range(2):
    b = h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop()
    p, n = h.paths()
    p = np.asarray(p).reshape(sc["n_agents"], -1, 3); n = np.asarray(n).reshape(-1)
    print("tick", k, "best", b, "n_points", n[:12], "nan agents", int(np.isnan(p[:, 1]).any(axis=1).sum()))
    for a in (0, 1, 5):
        print("  agent", a, "p1", p[a, 1], "p2", p[a, 2], "p3", p[a, 3])
h.close()

import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = pm.scenes.config_scene(name)
h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
h.set_initial_position(sc["start"])
b = h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"]); h.stop()
p, n = h.paths()
p = np.asarray(p).reshape(sc["n_agents"], -1, 3); n = np.asarray(n).reshape(-1)
print("n_points", n[:8])
for a in (0, 5):
    print("agent", a)
    for k in range(max(0, n[a] - 4), min(n[a] + 1, p.shape[1])):
        print("   ", k, p[a, k])
print("min_obs", np.asarray(h.min_obs_dist()).reshape(-1)[:6], "vel", np.asarray(h.agent_vel()).reshape(-1, 3)[:2])
print("known any", np.asarray(h.known()).any(), "rot nan", np.isnan(np.asarray(h.rot_vecs())).any())
h.close()

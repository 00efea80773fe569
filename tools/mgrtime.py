"""k_manager phase timing: run under rocprofv3 --kernel-trace --stats with mode = select | move | reset | tick"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
mode = sys.argv[1]
sc = pm.scenes.config_scene(sys.argv[2] if len(sys.argv) > 2 else "C2")
h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"])
for _ in range(3): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
h.stop()
for _ in range(200):
    if mode == "select":
        h.evaluate(sc["cost_gains"], sc["ws_limits"])
    elif mode == "move":
        h.move_real(sc["obstacles"], sc["dt"], 1, 0)
    elif mode == "reset":
        pos, vel, _ = h.real_state(); h.reset_agents(pos, vel, sc["obstacles"])
    else:
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"]); h.stop()
h.close()

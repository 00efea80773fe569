"""Where a C2 tick's time goes, from the device's own wall clock (needs a -DPMAF_TICK_STAMPS build:
PMAF_LIB_PATH=tools/dbg/stamps/libpmaf_hip.so python tools/tickstamps.py): every rollout wave's start / end and the
manager's start / end, absolute 100 MHz ticks, printed by the kernels; this script issues a few back-to-back ticks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
sc = pm.scenes.config_scene(sys.argv[1] if len(sys.argv) > 1 else "C2")
h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"])
for k in range(6):
    h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
h.stop(); h.close()

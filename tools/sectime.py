"""Section cycle counts of the w64 step (needs a -DPMAF_SECTION_TIMERS build:
PMAF_LIB_PATH=tools/dbg/libpmaf_hip_timers.so python tools/sectime.py C2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
for name in (sys.argv[1:] or ["C2"]):
    sc = pm.scenes.config_scene(name)
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    h.set_initial_position(sc["start"])
    for k in range(3):
        print("--- tick", k, flush=True)
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop()
    h.close()

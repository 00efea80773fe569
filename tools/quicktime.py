"""tick / rollout-kernel time of a BASELINE config per lanes-per-agent mapping (FAST=1: opt-in fast arithmetic).
usage: python tools/quicktime.py C2:64,32 C3:64 ..."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as g
pm=g.load_package()
import os
FAST = os.environ.get("FAST") == "1"
cfgs = sys.argv[1:] or ["C2:64,32", "C3:64,16", "C1:64"]
for c in cfgs:
    name,l=c.split(":"); lpas=[int(x) for x in l.split(",")]
    sc=pm.scenes.config_scene(name)
    for lpa in lpas:
        h=pm.PmafPlanner(sc,device=0,mgr_init_pos=sc["start"],lanes_per_agent=lpa, fast_math=FAST); h.set_initial_position(sc["start"])
        h.set_profiling(True)
        for _ in range(20): h.tick(None,sc["dt"],sc["cost_gains"],sc["ws_limits"])
        h.stop(); h.reset_kernel_stats()
        t0=time.perf_counter(); K=200
        for _ in range(K): h.tick(None,sc["dt"],sc["cost_gains"],sc["ws_limits"])
        h.stop(); t1=time.perf_counter()
        ms,n,steps=h.kernel_stats()
        print(name,"fast" if FAST else "strict","lpa",lpa,"tick %.1f us"%((t1-t0)/K*1e6),"kernel %.1f us"%(ms/n*1e3),"us/step %.3f"%(ms/n*1e3/(sc["max_prediction_steps"]-1)),"rollouts/s %.0f"%(sc["n_agents"]*K/(t1-t0)), flush=True)
        h.close()

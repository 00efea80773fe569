"""C5 (P populations x 1024 agents in one handle): tick and rollout-kernel time per lanes-per-agent mapping.
usage: python tools/c5time.py [P] [lpa,lpa,...]"""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as g
pm=g.load_package()
P=int(sys.argv[1]) if len(sys.argv)>1 else 8
lpas=[int(x) for x in (sys.argv[2] if len(sys.argv)>2 else "0").split(",")]
scs=[pm.scenes.config_scene("C5",scene_id=i) for i in range(P)]
starts=np.stack([s["start"] for s in scs]); sc=scs[0]
for lpa in lpas:
    h=pm.PmafPlanner(scs,device=0,mgr_init_pos=starts,lanes_per_agent=lpa); h.set_initial_position(starts)
    h.set_profiling(True)
    for _ in range(3): h.tick(None,sc["dt"],sc["cost_gains"],sc["ws_limits"])
    h.stop(); h.reset_kernel_stats()
    t0=time.perf_counter(); K=20
    for _ in range(K): h.tick(None,sc["dt"],sc["cost_gains"],sc["ws_limits"])
    h.stop(); t1=time.perf_counter()
    ms,n,steps=h.kernel_stats()
    print("C5 P",P,"cfg",h.launch_config(),"tick %.1f us"%((t1-t0)/K*1e6),"kernel %.1f us"%(ms/n*1e3),"rollouts/s %.0f"%(P*sc["n_agents"]*K/(t1-t0)),"agent-steps/s %.3g"%(steps/(t1-t0)), flush=True)
    h.close()

"""kernel time of the bench workload for scene ids 0..7 (what the ranks of a multi-GPU bench run would each see)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
for sid in range(8):
    sc = pm.scenes.config_scene(cfg, scene_id=sid)
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"]); h.set_initial_position(sc["start"]); h.set_profiling(True)
    for k in range(20): h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); h.reset_kernel_stats(); t0 = time.perf_counter(); K = 512
    for k in range(K):
        if k % 256 == 0: h.set_initial_position(sc["start"])
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop(); t1 = time.perf_counter(); ms, n, st = h.kernel_stats()
    print(cfg, "scene", sid, "tick %.1f us kernel %.1f us rollouts/s %.0f h_eff %.1f" % ((t1 - t0) / K * 1e6, ms / n * 1e3, sc["n_agents"] * K / (t1 - t0), st / n / sc["n_agents"]), flush=True)
    h.close()

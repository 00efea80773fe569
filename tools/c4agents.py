"""C4 (dual arm, 2 x 256 agents, trailing repulsive obstacle = the other arm's end effector): per-agent rollout
durations by heuristic type over an episode with host-side coupling. usage: python tools/c4agents.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
NAMES = {1: "GOAL", 2: "OBST", 3: "GOALOBST", 4: "VEL", 5: "RANDOM", 6: "HAD"}
scs = pm.scenes.dual_arm_scenes()
starts = np.stack([s["start"] for s in scs]); sc = scs[0]
h = pm.PmafPlanner(scs, device=0, mgr_init_pos=starts); h.set_initial_position(starts); h.set_profiling(True)
cpl = pm.shard.DualArmCoupling(np.stack([s["obstacles"] for s in scs]))
obs = cpl.coupled_obstacles(starts)
types = pm.scenes.default_agent_types(sc["n_agents"])
acc = []
for k in range(60):
    h.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]); h.stop()
    pos = np.asarray(h.real_state()[0]).reshape(2, 3)
    obs = cpl.coupled_obstacles(pos)
    if k >= 10: acc.append(np.asarray(h.prediction_times_ns()).reshape(2, -1))
t = np.mean(acc, axis=0) / 1e3
ms, n, st = h.kernel_stats()
print("C4 kernel %.1f us; per-agent rollout us: max %.1f median %.1f min %.1f" % (ms / n * 1e3, t.max(), np.median(t), t.min()))
for ty in sorted(set(types.tolist())):
    sel = t[:, types == ty]
    print("  %-9s n=%3d  mean %.1f  max %.1f" % (NAMES.get(ty, ty), sel.size, sel.mean(), sel.max()))
fl = t.reshape(-1); order = np.argsort(-fl)[:10]
print("  slowest:", [(int(i // t.shape[1]), int(i % t.shape[1]), NAMES.get(int(types[i % t.shape[1]])), round(float(fl[i]), 1)) for i in order])
h.close()

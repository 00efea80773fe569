"""C5: distribution of the per-agent (= per group of lanes) rollout durations and of
the per-WAVE duration (max over the wave's agents) in the group kernel.
usage: python tools/c5agents.py [lanes_per_agent] [populations] [config]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
lpa = int(sys.argv[1]) if len(sys.argv) > 1 else 0
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfgname = sys.argv[3] if len(sys.argv) > 3 else "C5"
scs = [pm.scenes.config_scene(cfgname, scene_id=i) for i in range(P)]
starts = np.stack([s["start"] for s in scs]); sc = scs[0]
h = pm.PmafPlanner(scs, device=0, mgr_init_pos=starts, lanes_per_agent=lpa); h.set_initial_position(starts)
cfg = h.launch_config()
acc = []
for k in range(12):
    h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"]); h.stop()
    if k >= 4: acc.append(np.asarray(h.prediction_times_ns()).reshape(P, -1))
t = np.mean(acc, axis=0) / 1e3
apw = 64 // cfg["lanes_per_agent"]
w = t.reshape(P, -1, apw).max(axis=2)
print("cfg", cfg)
print("agent us: min %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (t.min(), np.median(t), np.percentile(t, 90), np.percentile(t, 99), t.max()))
print("wave  us: min %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f  mean %.0f" % (w.min(), np.median(w), np.percentile(w, 90), np.percentile(w, 99), w.max(), w.mean()))
print("per population wave us (mean / max):", " ".join("%.0f/%.0f" % (w[i].mean(), w[i].max()) for i in range(P)))
wl = w.reshape(-1); half = 1024
print("launch order, per 1024 waves (mean / max):", " ".join("%.0f/%.0f" % (wl[i:i + half].mean(), wl[i:i + half].max()) for i in range(0, wl.size, half)))
print("per population, wave index quartiles (mean us):", [[int(q.mean()) for q in np.array_split(w[i], 4)] for i in range(P)])
types = pm.scenes.default_agent_types(sc["n_agents"])
for ty in sorted(set(types.tolist())):
    sel = t[:, types == ty]
    print("  type %d n=%d mean %.0f max %.0f" % (ty, sel.size, sel.mean(), sel.max()))
h.close()
# which waves are the slowest (population, wave index): the special heuristic types sit in wave 0 of every population
order = np.argsort(-w.reshape(-1))[:12]
print("slowest waves (population, wave, us):", [(int(i // w.shape[1]), int(i % w.shape[1]), round(float(w.reshape(-1)[i]), 1)) for i in order])
print("wave 0 of each population us:", [round(float(x), 1) for x in w[:, 0]])

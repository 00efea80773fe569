"""Rollout-kernel / tick time of the BASELINE configs per arithmetic policy on ONE box, interleaved (boxes differ by
+-2 %, runs on one box by +-0.1 %): strict (default, bit-exact), fast (PMAF_FLAG_FAST_MATH), contracted
(PMAF_FLAG_CONTRACTED). usage: python tools/policytime.py [C1 C2 C3 C4 C5] [--rounds 3] [--policies strict,contracted] [--out file.json]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
args = sys.argv[1:]
rounds, out = 3, None
if "--rounds" in args: k = args.index("--rounds"); rounds = int(args[k + 1]); del args[k:k + 2]
if "--out" in args: k = args.index("--out"); out = args[k + 1]; del args[k:k + 2]
only = None
if "--policies" in args: k = args.index("--policies"); only = args[k + 1].split(","); del args[k:k + 2]
cfgs = args or ["C1", "C2", "C3", "C4", "C5"]
POL = {"strict": {}, "fast": {"fast_math": True}, "contracted": {"contracted": True}}
if only: POL = {k: v for k, v in POL.items() if k in only}

def scenes_of(c):
    if c == "C4": return pm.scenes.dual_arm_scenes()
    if c == "C5": return [pm.scenes.config_scene("C5", scene_id=i) for i in range(8)]
    return [pm.scenes.config_scene(c)]

res = {}
for c in cfgs:
    scs = scenes_of(c); sc = scs[0]; P = len(scs)
    starts = np.stack([s["start"] for s in scs])
    obs = np.stack([s["obstacles"] for s in scs])
    K = {"C1": 400, "C2": 300, "C3": 60, "C4": 200, "C5": 40}[c]
    for r in range(rounds):
        for pol, kw in POL.items():
            h = pm.PmafPlanner(scs if P > 1 else sc, device=0, mgr_init_pos=starts if P > 1 else starts[0], **kw)
            h.set_initial_position(starts if P > 1 else starts[0])
            h.set_profiling(True)
            # C4: each arm's trailing obstacle follows the other arm's end effector (host-coupled here; the kernel time is
            # what is compared)
            cpl = pm.shard.DualArmCoupling(obs, 0.1) if c == "C4" else None
            live = lambda: cpl.coupled_obstacles(h.real_state()[0]) if cpl is not None else None
            for _ in range(5): h.tick(live(), sc["dt"], sc["cost_gains"], sc["ws_limits"])
            h.stop(); h.reset_kernel_stats()
            t0 = time.perf_counter()
            for _ in range(K): h.tick(live(), sc["dt"], sc["cost_gains"], sc["ws_limits"])
            h.stop(); t1 = time.perf_counter()
            ms, n, steps = h.kernel_stats()
            rec = dict(tick_us=(t1 - t0) / K * 1e6, kernel_us=ms / n * 1e3, rollouts_per_s=P * sc["n_agents"] * K / (t1 - t0),
                       h_eff=steps / (n * P * sc["n_agents"]))
            res.setdefault(c, {}).setdefault(pol, []).append(rec)
            print(c, pol, "round", r, "tick %.1f us kernel %.1f us rollouts/s %.0f h_eff %.1f" %
                  (rec["tick_us"], rec["kernel_us"], rec["rollouts_per_s"], rec["h_eff"]), flush=True)
            h.close()
summary = {c: {p: dict(kernel_us=float(np.median([x["kernel_us"] for x in v])), tick_us=float(np.median([x["tick_us"] for x in v])))
               for p, v in d.items()} for c, d in res.items()}
print(json.dumps(summary, indent=1))
if out: json.dump(dict(summary=summary, runs=res), open(out, "w"), indent=1)

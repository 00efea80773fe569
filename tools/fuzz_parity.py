"""randomised parity run: random scenes (agent / obstacle counts, heuristic mixes, moving obstacles, gains, horizons,
lanes-per-agent mappings; PMAF_FUZZ_MANY=1: 900 ... 4 600 agents per population, the mappings the measured table chooses) through the HIP path and the CPU oracle (portable-exp mode), every result compared bit for bit.
usage: python tools/fuzz_parity.py [n_trials] [seed] [only_trial]   (only_trial: replay the generator, run and
diff that one trial in detail)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
from oracle import orc
orc.set_exp_mode(1)
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
MANY = os.environ.get("PMAF_FUZZ_MANY") == "1"
bad = 0
kinds = {}   # (lanes per agent, waves per agent, priority slices) -> trials
t0 = time.time()
for trial in range(n_trials):
    N = int(rng.integers(1, 40)); M = int(rng.choice([0, 1, 2, 5, 9, 17, 32, 33, 58, 59, 60, 61, 62, 63, 64, 65, 100, 122, 123, 128, 129, 150, 192, 200, 256]))
    H = int(rng.integers(5, 160)); dyn = bool(rng.integers(0, 2))
    if rng.integers(0, 25) == 0:   # a large population (multi-block launches of the group kernels), short horizon
        N = int(rng.integers(200, 1500)); H = int(rng.integers(5, 30)); M = int(rng.choice([3, 17, 32, 40]))
    if MANY:   # PMAF_FUZZ_MANY=1 (round 6): every trial is a many-agent population -- the shapes pick_lpa's measured table decides
        # (16 / 32 / 64 lanes per agent, the wave per agent's priority-slicing loop between 1 025 and 2 048 agents), short horizons
        N = int(rng.choice([rng.integers(1025, 2049), rng.integers(2049, 4600), rng.integers(900, 1100)]))
        H = int(rng.integers(4, 22)); M = int(rng.choice([0, 3, 9, 16, 17, 32, 33, 48, 59, 60, 61, 64, 65, 100]))
    types = rng.integers(1, 7, N).astype(np.int32) if rng.integers(0, 2) else None
    sc = pm.scenes.synthetic_scene(N, H, M, 11, trial, dynamic=dyn, agent_types=types)
    if rng.integers(0, 3) == 0:   # denser clutter around the path
        k = min(M, 6)
        sc["obstacles"][:k, :3] = np.c_[rng.uniform(-0.5, 0.5, k), rng.uniform(-0.08, 0.08, k), 0.7 + rng.uniform(-0.08, 0.08, k)]
    if rng.integers(0, 4) == 0:   # the repulsive obstacle near the path
        sc["obstacles"][-1] = [rng.uniform(-0.3, 0.3), rng.uniform(0.1, 0.4), 0.7, 0, rng.uniform(-0.3, 0.0), 0, 0.1]
    sc["k_circ"] = float(rng.choice([0.025, 0.015, 0.05])); sc["k_damp"] = float(rng.choice([3.0, 4.0]))
    wild = rng.integers(0, 3) == 0   # every third trial: parameters well off the shipped task files
    if wild:
        sc["dt"] = float(rng.choice([0.01, 0.02, 0.005, 0.05])); sc["velocity_max"] = float(rng.choice([0.2, 0.5, 1.0, 0.05]))
        sc["approach_dist"] = float(rng.choice([0.25, 0.1, 0.6])); sc["detect_shell_rad"] = float(rng.choice([0.35, 0.8, 0.1, 0.05]))
        sc["agent_mass"] = float(rng.choice([1.0, 1.0, 0.5, 2.5])); sc["radius"] = float(rng.choice([0.05, 0.0, 0.12]))
        sc["k_attr"] = float(rng.choice([4.0, 0.0, 1.0, 20.0])); sc["k_repel"] = float(rng.choice([0.08, 0.0, 1.0]))
        sc["k_circ"] = float(rng.choice([0.025, 0.0, 0.5, 5.0]))
        sc["cost_gains"] = np.array([rng.choice([100.0, 1.0, 0.0]), rng.choice([10.0, 0.0, 50.0]), rng.choice([0.001, 1.0, 0.0]), rng.choice([1.0, 10.0, 0.0])])
        sc["ws_limits"] = np.array([rng.uniform(0.0, 1.0), rng.uniform(-1.0, 0.0), rng.uniform(0.0, 0.3), rng.uniform(-0.3, 0.0), rng.uniform(0.7, 1.1), rng.uniform(0.2, 0.7)])
        if rng.integers(0, 3) == 0: sc["start"] = sc["goal"] + rng.uniform(-0.3, 0.3, 3)          # short trips, early goal
        if rng.integers(0, 3) == 0 and M > 0: sc["start"] = sc["obstacles"][0, :3] + rng.uniform(-0.02, 0.02, 3)  # inside an obstacle
        if rng.integers(0, 4) == 0 and M > 2: sc["obstacles"][1, :3] = sc["obstacles"][0, :3]       # coincident obstacles
    lpa = int(rng.choice([0, 0, 64, 32, 16, 8, 4, 1]))
    if MANY and rng.integers(0, 4): lpa = 0   # mostly the library's own choice
    if lpa and (M + lpa - 1) // max(lpa, 1) > 64: lpa = 0
    ticks = int(rng.integers(1, 6)) if rng.integers(0, 8) else int(rng.integers(10, 40))
    if MANY: ticks = min(ticks, 4)
    ieee = bool(rng.integers(0, 6) == 0)
    # 62..256 obstacles: the split kernel with the fewest waves (default), with more waves than needed, or the one-wave kernels
    os.environ["PMAF_MW"] = str(rng.choice(["", "", "0", "3", "4"]))
    if only >= 0 and trial != only: continue
    try:
        hip = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"], lanes_per_agent=lpa, ieee_sequences=ieee)
    except pm.PmafError as e:
        print("trial", trial, "create refused:", e); continue
    cfg_ = hip.launch_config(); kinds[(cfg_["lanes_per_agent"], cfg_["waves_per_agent"], cfg_.get("priority_slices", False))] = kinds.get((cfg_["lanes_per_agent"], cfg_["waves_per_agent"], cfg_.get("priority_slices", False)), 0) + 1
    ora = orc.OraclePlanner(sc, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"]); ora.set_initial_position(sc["start"])
    obs = sc["obstacles"].copy(); ok = True
    for t in range(ticks):
        bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]); bo = ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        ok &= bool(np.all(np.asarray(bh) == np.asarray(bo)))
        if dyn: obs = pm.scenes.advance_live_obstacles(obs)
    hip.stop()
    ph, nh = hip.paths(); po, no = ora.paths()
    same = lambda a, b: np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    ok &= same(nh, no) and same(ph, po) and same(hip.costs(), ora.costs()) and same(hip.min_obs_dist(), ora.min_obs_dist())
    ok &= same(hip.known(), ora.known()) and same(hip.rot_vecs(), ora.rot_vecs()) and same(hip.path_lengths(), ora.path_lengths())
    for a, b in zip(hip.real_state(), ora.real_state()): ok &= same(a, b)
    if only >= 0:
        print(dict(N=N, M=M, H=H, dyn=dyn, lpa=lpa, ticks=ticks, types=None if types is None else types.tolist()))
        for name, a, b in (("n", nh, no), ("paths", ph, po), ("costs", hip.costs(), ora.costs()), ("min_obs", hip.min_obs_dist(), ora.min_obs_dist()),
                           ("known", hip.known(), ora.known()), ("rot", hip.rot_vecs(), ora.rot_vecs()), ("len", hip.path_lengths(), ora.path_lengths())):
            a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
            d = ~((a == b) | (np.isnan(a) & np.isnan(b)))
            print(name, "mismatching entries", int(d.sum()), "of", a.size, "first", np.argwhere(d)[:4].tolist(), a[d][:4], b[d][:4])
        rh, ro_ = np.asarray(hip.rot_vecs()), np.asarray(ora.rot_vecs())
        dd = np.argwhere(~((rh == ro_) | (np.isnan(rh) & np.isnan(ro_))))
        for ix in dd[:3]:
            ag, ob = int(ix[0]), int(ix[1])
            print("rot agent", ag, "obstacle", ob, "hip", [float(x).hex() for x in rh[ag, ob]], "oracle", [float(x).hex() for x in ro_[ag, ob]])
        print("obstacles", sc["obstacles"]); print("types", pm.scenes.default_agent_types(N) if types is None else types)
    if not ok:
        bad += 1
        print("MISMATCH trial", trial, dict(N=N, M=M, H=H, dyn=dyn, lpa=lpa, ticks=ticks, wild=bool(wild), ieee=ieee, cfg=hip.launch_config()), flush=True)
    hip.close(); ora.close()
print("kernels run (lanes per agent, waves per agent, priority slices): trials", dict(sorted(kinds.items())))
print("trials", n_trials, "mismatches", bad, "in %.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)

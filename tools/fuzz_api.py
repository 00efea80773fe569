"""randomised API-level parity run: P = 1..5 populations in one handle (independent scenes), each tick issued either
as pmaf_tick or as the reference's five-call sequence (stop / evaluate / move_real / reset_agents / start), with
save_state -> new handle -> load_state hand-overs in between; every population compared bit for bit with its own
CPU oracle. usage: python tools/fuzz_api.py [n_trials] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
from oracle import orc
orc.set_exp_mode(1)
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
same = lambda a, b: np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
bad = 0
t0 = time.time()
for trial in range(n_trials):
    P = int(rng.integers(1, 6)); N = int(rng.integers(2, 40)); M = int(rng.choice([1, 3, 9, 32, 40, 70, 130])); H = int(rng.integers(10, 120))
    dyn = bool(rng.integers(0, 2)); lpa = int(rng.choice([0, 0, 64, 32, 16, 8]))
    types = rng.integers(1, 7, N).astype(np.int32) if rng.integers(0, 2) else None
    scs = []
    for p in range(P):
        sc = pm.scenes.synthetic_scene(N, H, M, 12, trial * 8 + p, dynamic=dyn, agent_types=types)
        sc["goal"] = sc["goal"] + rng.uniform(-0.1, 0.1, 3); sc["start"] = sc["start"] + rng.uniform(-0.05, 0.05, 3)
        scs.append(sc)
    sc0 = scs[0]
    starts = np.stack([s["start"] for s in scs])
    mk = lambda: pm.PmafPlanner(scs, device=0, mgr_init_pos=starts, lanes_per_agent=lpa)
    try:
        hip = mk()
    except pm.PmafError as e:
        print("trial", trial, "create refused:", e); continue
    oras = [orc.OraclePlanner(s, mgr_init_pos=s["start"]) for s in scs]
    hip.set_initial_position(starts)
    for o, s in zip(oras, scs): o.set_initial_position(s["start"])
    obs = np.stack([s["obstacles"] for s in scs]); ok = True
    for t in range(int(rng.integers(2, 10))):
        style = int(rng.integers(0, 3))
        if style == 0:
            bh = np.asarray(hip.tick(obs, sc0["dt"], sc0["cost_gains"], sc0["ws_limits"])).reshape(-1)
        else:
            hip.stop()
            bh = np.asarray(hip.evaluate(sc0["cost_gains"], sc0["ws_limits"])).reshape(-1)
            hip.move_real(obs, sc0["dt"], 1, bh.astype(np.int32))
            pos, vel, _ = hip.real_state()
            hip.reset_agents(pos, vel, obs)
            hip.start()
        bo = np.array([o.tick(obs[p], sc0["dt"], sc0["cost_gains"], sc0["ws_limits"]) for p, o in enumerate(oras)])
        ok &= same(bh, bo)
        if dyn: obs = np.stack([pm.scenes.advance_live_obstacles(o) for o in obs])
        if rng.integers(0, 5) == 0:   # hand the planner over to a fresh handle through a state blob
            blob = hip.save_state(); hip.close(); hip = mk(); hip.load_state(blob)
    hip.stop()
    ph, nh = hip.paths()
    ph = np.asarray(ph).reshape(P, N, -1, 3); nh = np.asarray(nh).reshape(P, N)
    costs = np.asarray(hip.costs()).reshape(P, N); rots = np.asarray(hip.rot_vecs()).reshape(P, N, -1, 3)
    rp = [np.asarray(x).reshape(P, 3) for x in hip.real_state()]
    for p, o in enumerate(oras):
        po, no = o.paths()
        ok &= same(nh[p], no) and same(ph[p], po) and same(costs[p], o.costs()) and same(rots[p], o.rot_vecs())
        for a, b in zip(rp, o.real_state()): ok &= same(a[p], np.asarray(b).reshape(-1))
        o.close()
    if not ok:
        bad += 1
        print("MISMATCH trial", trial, dict(P=P, N=N, M=M, H=H, dyn=dyn, lpa=lpa, cfg=hip.launch_config()), flush=True)
    hip.close()
print("trials", n_trials, "mismatches", bad, "in %.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)

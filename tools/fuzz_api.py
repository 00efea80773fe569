"""randomised API-level parity run: P = 1..5 populations in one handle (independent scenes), each tick issued either
as pmaf_tick or as the reference's five-call sequence (stop / evaluate / move_real / reset_agents / start), with
save_state -> new handle -> load_state hand-overs in between; every population compared bit for bit with its own
CPU oracle. Round 5 also draws the node's other boundary paths: closed loop (pmaf_set_real_position with a tracking
error in front of every tick, B/src/panda_bimanual_control.cpp:333-335), prediction_freq_multiple in {1, 2, 3} (rollout
dt = multiple x the real step's, B/src/cf_manager.cpp:118-123) and a goal change mid-run (new handle towards a new goal
with fresh Random vectors, the best agent carried over by pmaf_set_best, B/src/cf_manager.cpp:344-354).
usage: python tools/fuzz_api.py [n_trials] [seed]"""
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
    closed = bool(rng.integers(0, 3) == 0); lag = float(rng.choice([0.3, 0.05, 1.0])); mult = int(rng.choice([1, 1, 2, 3]))
    regoal = bool(rng.integers(0, 3) == 0)
    scs = []
    for p in range(P):
        sc = pm.scenes.synthetic_scene(N, H, M, 12, trial * 8 + p, dynamic=dyn, agent_types=types)
        sc["goal"] = sc["goal"] + rng.uniform(-0.1, 0.1, 3); sc["start"] = sc["start"] + rng.uniform(-0.05, 0.05, 3)
        dt_real = sc["dt"]; sc["dt"] = mult * dt_real
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
    measured = starts.copy()
    n_ticks = int(rng.integers(2, 10)); regoal_at = int(rng.integers(1, n_ticks)) if regoal else -1
    for t in range(n_ticks):
        if t == regoal_at:   # taskCallback's PLAN branch: a new population towards a new goal, best agent carried over
            hip.stop()
            cur = np.asarray(hip.real_state()[0]).reshape(P, 3).copy()
            bt, bi = [np.asarray(x).copy() for x in hip.best()]
            old_rv = [s["random_vecs"] for s in scs]
            for p, s in enumerate(scs):
                s["goal"] = s["goal"] + rng.uniform(-0.3, 0.3, 3); s["obstacles"] = obs[p].copy()
                s["random_vecs"] = pm.scenes.synthetic_scene(N, H, M, 13, trial * 8 + p, agent_types=types)["random_vecs"]
            carry = np.stack([old_rv[p][max(int(bi[p]), 1) - 1] for p in range(P)])
            hip.close(); hip = pm.PmafPlanner(scs, device=0, mgr_init_pos=cur, lanes_per_agent=lpa)
            new_oras = [orc.OraclePlanner(s, mgr_init_pos=cur[p]) for p, s in enumerate(scs)]
            if (bi > 0).all():
                hip.set_best(bi, bt, carry)
                for p, o in enumerate(new_oras): o.set_best(int(bi[p]), int(bt[p]), carry[p])
            for o in oras: o.close()
            oras = new_oras
            hip.set_initial_position(cur)
            for p, o in enumerate(oras): o.set_initial_position(cur[p])
            measured = cur.copy()
        if closed:
            hip.set_real_position(measured)
            for p, o in enumerate(oras): o.set_real_position(measured[p])
        style = int(rng.integers(0, 3))
        if style == 0:
            bh = np.asarray(hip.tick(obs, dt_real, sc0["cost_gains"], sc0["ws_limits"])).reshape(-1)
        else:
            hip.stop()
            bh = np.asarray(hip.evaluate(sc0["cost_gains"], sc0["ws_limits"])).reshape(-1)
            hip.move_real(obs, dt_real, 1, bh.astype(np.int32))
            pos, vel, _ = hip.real_state()
            hip.reset_agents(pos, vel, obs)
            hip.start()
        bo = np.array([o.tick(obs[p], dt_real, sc0["cost_gains"], sc0["ws_limits"]) for p, o in enumerate(oras)])
        ok &= same(bh, bo)
        sp = np.asarray(hip.real_state()[0]).reshape(P, 3)
        measured = sp - lag * (sp - measured)
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
        ok &= same(hip.real_path(p), o.real_path())
        o.close()
    if not ok:
        bad += 1
        print("MISMATCH trial", trial, dict(P=P, N=N, M=M, H=H, dyn=dyn, lpa=lpa, closed=closed, lag=lag, mult=mult, regoal_at=regoal_at, cfg=hip.launch_config()), flush=True)
    hip.close()
print("trials", n_trials, "mismatches", bad, "in %.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)

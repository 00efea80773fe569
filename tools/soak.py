"""Soak run of the tick protocol: ONE handle ticked a very large number of times in closed loop (the measured position
handed over in front of every tick) with a NEW obstacle list on every tick (zero-copy pinned hand-over), the real agent
put back every `period` ticks -- against the CPU oracle (TEST INFRASTRUCTURE: the checker) in lock step: best index,
set-point and velocity compared bit for bit on EVERY tick, all agents' paths and costs every 1000th. What a short test
cannot see: a hand-over buffer reused one tick too early, the sequence number at 2^20 and beyond, a mailbox read torn
once in a million ticks.
usage: python tools/soak.py [n_ticks = 200000] [config = C1] [period = 150]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
from oracle import orc
orc.set_exp_mode(1)
n_ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
cfg = sys.argv[2] if len(sys.argv) > 2 else "C1"
period = int(sys.argv[3]) if len(sys.argv) > 3 else 150
sc = pm.scenes.config_scene(cfg, dynamic=True)
hip = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
ora = orc.OraclePlanner(sc, mgr_init_pos=sc["start"])
same = lambda a, b: np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
bad, t0, last = 0, time.time(), time.time()
obs = sc["obstacles"].copy()
meas = np.asarray(sc["start"], dtype=np.float64).copy()
for t in range(n_ticks):
    if t % period == 0:
        hip.set_initial_position(sc["start"]); ora.set_initial_position(sc["start"])
        obs = sc["obstacles"].copy(); meas = np.asarray(sc["start"], dtype=np.float64).copy()
    hip.set_real_position(meas); ora.set_real_position(meas)
    bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    bo = ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    ph, vh = hip.real_state()[0], hip.real_state()[1]
    po, vo = ora.real_state()[0], ora.real_state()[1]
    ok = bh == bo and same(ph, po) and same(vh, vo)
    if ok and t % 1000 == 999:
        hip.stop()
        ok = same(hip.paths()[0], ora.paths()[0]) and same(hip.paths()[1], ora.paths()[1]) and same(hip.costs(), ora.costs())
    if not ok:
        bad += 1
        if bad <= 5: print("MISMATCH at tick", t, bh, bo, ph, po, flush=True)
    sp = np.asarray(ph, dtype=np.float64).reshape(3)
    meas = sp - 0.3 * (sp - meas)          # a controller that lags 30 % behind the set-point
    obs = pm.scenes.advance_live_obstacles(obs)
    if time.time() - last > 60:
        last = time.time(); print("  ... tick %d, %d mismatches, %.0f s" % (t + 1, bad, time.time() - t0), flush=True)
hip.stop()
print("soak %s: %d closed-loop ticks with a new obstacle list each, real agent put back every %d: %d mismatches in %.0f s"
      % (cfg, n_ticks, period, bad, time.time() - t0))
hip.close(); ora.close()
sys.exit(1 if bad else 0)

"""Tolerance-parity probe (uses tests/test_tolerance_gpu.py's lock-step comparison): which kernel / scene produces the
deviations of a non-bit-exact policy. usage: python tools/tolprobe.py <case> ... with case =
config:scene_id:n_agents:lanes_per_agent:policy:ticks  (n_agents 0 = the config's own)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pm = g.load_package()
from oracle import orc
orc.build()
import test_tolerance_gpu as T
orc.set_exp_mode(0)
for case in sys.argv[1:]:
    cfg, sid, n, lpa, pol, ticks = case.split(":")
    sc = pm.scenes.config_scene(cfg, scene_id=int(sid))
    if int(n):
        c = pm.scenes.CONFIGS[cfg]
        sc = pm.scenes.synthetic_scene(int(n), c["horizon"], c["n_field"], {"C2": 2, "C3": 3, "C5": 5}[cfg], int(sid))
    kw = dict(T.POLICIES[pol]) if pol in T.POLICIES else {"fast_math": True}
    hip, oras = T._build(pm, orc, [sc], lanes_per_agent=int(lpa), **kw)
    st = T.lockstep(hip, oras, [sc], int(ticks))
    T.report(case, pol, st)
    hip.close()

"""How much do the two details of the reference's arithmetic that cannot be pinned here (the
association of Eigen's 3-vector dot product; vector / scalar as a true division) move the planner's output?

The oracle is rebuilt with the alternatives and run closed-loop on BASELINE C1 / C2: the real agent's set-points and all agents'
predicted paths (100- and 200-step rollouts through the obstacles) stay within 1e-9 m of the default build's -- four orders of magnitude inside the north-star
tolerance (1e-5 m) the GPU path is held to. So whichever way the real Eigen evaluates these, the tolerance claim does
not hinge on it. (Bit-exact claims are against THIS oracle only)"""
import os
import subprocess
import sys

import numpy as np
import pytest

import conftest

SCRIPT = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as g
pm = g.load_package()
from oracle import orc
orc.set_exp_mode(0)
out = {}
for name, ticks in (("C1", 40), ("C2", 12), ("C3", 3)):
    sc = pm.scenes.config_scene(name)
    o = orc.OraclePlanner(sc, mgr_init_pos=sc["start"]); o.set_initial_position(sc["start"])
    pos, best = [], []
    for _ in range(ticks):
        best.append(int(o.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])))
        pos.append(np.asarray(o.real_state()[0]).tolist())
    paths, n = o.paths()
    out[name] = dict(pos=pos, best=best, paths=np.asarray(paths).tolist(), n=np.asarray(n).tolist())
print(json.dumps(out))
"""


def _run(tmp_path, tag, defines):
    so = os.path.join(str(tmp_path), "liborc_%s.so" % tag)
    src = os.path.join(conftest.ROOT, "oracle", "pmaf_oracle.c")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math"] + defines +
                          ["-shared", "-o", so, src, "-lm", "-lpthread"])
    env = dict(os.environ, PMAF_ORACLE_LIB=so)
    r = subprocess.run([sys.executable, "-c", SCRIPT % conftest.ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_unpinned_eigen_details_move_the_set_points_by_less_than_a_nanometre(tmp_path):
    base = _run(tmp_path, "default", [])
    for tag, defs in (("rassoc", ["-DPMAF_DOT_RIGHT_ASSOC"]), ("recip", ["-DPMAF_QUOTIENT_BY_RECIPROCAL"]),
                      ("both", ["-DPMAF_DOT_RIGHT_ASSOC", "-DPMAF_QUOTIENT_BY_RECIPROCAL"])):
        alt = _run(tmp_path, tag, defs)
        for cfg in ("C1", "C2", "C3"):
            d = np.abs(np.asarray(alt[cfg]["pos"]) - np.asarray(base[cfg]["pos"])).max()
            assert alt[cfg]["n"] == base[cfg]["n"]
            dp = np.abs(np.asarray(alt[cfg]["paths"]) - np.asarray(base[cfg]["paths"])).max()   # all agents' last rollouts
            b = base[cfg]["best"][-1]   # the selected agent's predicted trajectory: what the north star's tolerance is about
            db = np.abs(np.asarray(alt[cfg]["paths"])[b] - np.asarray(base[cfg]["paths"])[b]).max()
            assert db < 1e-9, (tag, cfg, db)
            print(tag, cfg, "max deviation: set-points %.3e m, selected trajectory %.3e m, all predicted paths %.3e m" % (d, db, dp),
                  "same best sequence:", alt[cfg]["best"] == base[cfg]["best"])
            # C3's 500-step rollouts through 128 obstacles amplify a last-bit difference in a few non-selected agents
            # (DESIGN.md section 2); the selected trajectory / set-points are what the tolerance is about
            assert d < 1e-9 and (dp < 1e-9 or cfg == "C3"), (tag, cfg, d, dp)


def test_shipped_task_scenes_conditioning_under_the_unpinned_eigen_details():
    """The same question on the reference's OWN operating point: the nine shipped task scenes, closed loop until reached /
    900 ticks (tools/oracle_conditioning.py, record: profiles/r4_oracle_conditioning.txt). The six dual_arms_* scenes are
    well conditioned -- same best-index sequence, set-points within 1e-9 m whichever way the dot product associates and
    the quotient is formed. The three sim_kobo_dyn_spheres* scenes (H = 1500 / 1200 through moving spheres) are NOT: two
    IEEE-conformant evaluation orders of the same algorithm part ways there (set-points by centimetres, a best-index
    difference at tick 1 of spheres3), so against a reference whose evaluation order cannot be pinned no implementation
    can promise 1e-5 m on them -- bit-exact parity is against THIS oracle's order. This test pins
    which scenes are which."""
    r = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "tools", "oracle_conditioning.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    rows = {}
    variant = None
    for line in r.stdout.splitlines():
        if line.startswith("== variant:"):
            variant = line.split(":", 1)[1].strip()
        m = re.match(r"(\S+)\s+(\d+) ticks \| first best-index difference: (\S+)\s*\| set-point (\S+) m \| selected trajectory (\S+) m", line)
        if m and variant:
            rows[(variant, m.group(1))] = (int(m.group(2)), m.group(3), float(m.group(4)), float(m.group(5)))
    assert len(rows) == 27
    for (variant, task), (ticks, flip, dset, dsel) in rows.items():
        if task.startswith("dual_arms_"):
            assert flip == "None" and dset < 1e-9, (variant, task, flip, dset)
    chaotic = {task for (variant, task), (ticks, flip, dset, dsel) in rows.items() if flip != "None" or dset > 1e-5}
    print("task scenes on which the oracle's own evaluation-order variants part ways:", sorted(chaotic))
    assert chaotic <= {"sim_kobo_dyn_spheres1", "sim_kobo_dyn_spheres2", "sim_kobo_dyn_spheres3"}

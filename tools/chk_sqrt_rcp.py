"""bitwise check of a / sqrt(b) through the shared-reciprocal sequence (debug op 9) against IEEE"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
rng = np.random.default_rng(11)
bad = 0; tot = 0
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    n = 4_000_000
    if rnd % 3 == 0:
        a = rng.uniform(1e-12, 50.0, n) * rng.choice([-1.0, 1.0], n); b = rng.uniform(1e-12, 50.0, n)
    elif rnd % 3 == 1:
        a = np.ldexp(rng.uniform(1.0, 2.0, n), rng.integers(-200, 200, n)) * rng.choice([-1.0, 1.0], n)
        b = np.ldexp(rng.uniform(1.0, 2.0, n), rng.integers(-200, 200, n))
    else:  # norms of small vectors, as on the path
        v = rng.uniform(-1.5, 1.5, (n, 3)); b = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]; a = v[:, rnd % 3]
    with np.errstate(all="ignore"):
        ref = a / np.sqrt(b)
    got = pm.debug_math(9, a, b)
    m = got != ref
    bad += int(m.sum()); tot += n
    if m.any():
        i = np.flatnonzero(m)[:5]
        print("mismatch", a[i], b[i], got[i], ref[i], (got[i] - ref[i]) / np.spacing(ref[i]))
print("checked", tot, "mismatches", bad)

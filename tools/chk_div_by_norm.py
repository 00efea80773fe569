"""bitwise check of a / sqrt(b) through the kernels' sqrt + shared refined reciprocal (debug op 9) against IEEE,
random operands and the norms next to 1.0 that normalising cross products of unit vectors produces"""
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
# norms within a few ulps of 1 (and of other powers of two), numerators on rounding ties
k = np.arange(-64, 65, dtype=np.float64)
for base in (1.0, 0.5, 2.0, 0.25):
    sN = base * (1.0 + k * 2.0 ** -53)
    zN = sN * sN
    for num in (2.0 ** -55, 3.0 * 2.0 ** -55, 1.0, 0.3, 2.0 ** -30, 1.0 + 2.0 ** -52, 5.0 * 2.0 ** -60):
        for sg in (1.0, -1.0):
            a = np.full_like(zN, sg * num)
            with np.errstate(all="ignore"):
                ref = a / np.sqrt(zN)
            got = pm.debug_math(9, a, zN)
            m = got != ref
            bad += int(m.sum()); tot += a.size
            if m.any():
                print("near-one mismatch base", base, "num", sg * num, "k", k[m][:6])
print("checked", tot, "mismatches", bad)

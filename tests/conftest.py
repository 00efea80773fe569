import os
import sys

import numpy as np
import pytest

# PyTorch-ROCm wheels bundle their own HIP/HSA runtime; a process that loads
# libpmaf_hip.so (system ROCm) first and torch afterwards ends up with two HSA
# runtimes and torch reports "No HIP GPUs are available". Importing torch first
# makes both share one runtime (tests that own device buffers through torch
# need this; the product library itself never needs torch).
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pmaf():
    """the package predictive-multi-agent-framework_amd/ as module pmaf_amd"""
    return graft.load_package()


@pytest.fixture(scope="session")
def scenes(pmaf):
    return pmaf.scenes


@pytest.fixture(scope="session")
def oracle(pmaf):
    """the CPU oracle (checker). PMAF_VARIANT selects the evaluation-order pair: the library out of lib_<variant>/ and the
    oracle built with the same switch -- a 0-tolerance comparison across the two orders would be meaningless, so it is
    refused here"""
    from oracle import orc
    orc.build()
    try:
        L = pmaf.load_library()
    except OSError:
        L = None
    if L is not None:
        assert L.pmaf_eval_order() == orc.eval_order(), "library and oracle were built with different dot-product associations"
    return orc


# Tolerance tests of the CONTRACTED policy against the oracle's libm-exp mode rest on last-bit differences NOT being
# amplified past 1e-5 m. On chaotic rollouts (long horizons through moving spheres, C3's 500-step chains) whether they are
# depends on the evaluation order AND on every other last bit: with the round-4 exp the two cases below exceeded the
# tolerance under the right-associated pair (profiles/r6_gpu_tests_rassoc.log of that build); with glibc's exp they happen
# to hold (6 xpassed -> 2). Kept as NON-strict expected failures there: per-scene luck, not a contract. The strict kernels'
# comparisons (libm:* / task_libm:* / strict_libm:*) are exact under either association since round 5 and are not listed.
CHAOTIC_VS_LIBM = {
    "rassoc": {"contracted:C3", "contracted_task:sim_kobo_dyn_spheres2"},
}


_LIBM_SAME = {}


def libm_is_restated(orc):
    """True when the host's libm exp is the algorithm oracle/pmaf_oracle.c:pmaf_portable_exp (and the kernels) restate --
    glibc >= 2.28's FMA variant, as on the build image and the GPU boxes. Decided by comparing 200 000 results."""
    if "v" not in _LIBM_SAME:
        rng = np.random.default_rng(77)
        x = np.concatenate([-rng.uniform(0, 3, 120000), -rng.uniform(0, 500, 60000), rng.uniform(0, 709.79, 20000), rng.uniform(709.7, 709.79, 2000)])
        _LIBM_SAME["v"] = bool((orc.portable_exp(x) == orc.libm_exp(x)).all())
    return _LIBM_SAME["v"]


def libm_tol(orc, tol):
    """the tolerance of a strict-kernels-vs-libm-oracle comparison: 0 where the host libm is the restated algorithm
    (the kernels then ARE the reference's arithmetic, exp included), the north star's 1e-5 m elsewhere"""
    return 0.0 if libm_is_restated(orc) else tol


def expect_chaotic(request, case):
    """mark the running test as an expected failure when `case` is known to exceed the libm tolerance under the
    evaluation-order variant being tested (non-strict: a pass is fine)"""
    v = os.environ.get("PMAF_VARIANT", "")
    if case in CHAOTIC_VS_LIBM.get(v, ()):
        request.applymarker(pytest.mark.xfail(reason="chaotic rollouts amplify exp's last bit past 1e-5 m under PMAF_VARIANT=%s (%s)" % (v, case),
                                              strict=False))


def exe(path):
    """the C++ test driver built for the library variant under test (__graft_entry__.build: tools/plan_task[_rassoc],
    tests/cpp/facade_tick[_rassoc])"""
    v = os.environ.get("PMAF_VARIANT", "")
    return path + ("_" + v if v else "")


def binary_env(pmaf):
    """environment for the C++ test drivers (tools/plan_task, tests/cpp/facade_tick): their RUNPATH points at lib/, so
    the directory of the library under test goes in front of it (PMAF_VARIANT / PMAF_LIB_PATH builds)"""
    d = os.path.dirname(os.path.abspath(pmaf.LIB_PATH))
    old = os.environ.get("LD_LIBRARY_PATH", "")
    return dict(os.environ, LD_LIBRARY_PATH=d + (":" + old if old else ""))


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def hip_lib(pmaf):
    """built libpmaf_hip.so (built on demand; hipcc cross-compiles on CPU)"""
    graft.build()
    return pmaf.load_library()


def std_mt19937_unit_vectors(seed, n):
    """n normalised vectors from std::mt19937(seed) +
    std::uniform_real_distribution<>(-1,1) as libstdc++ evaluates them (two
    32-bit draws per double, low word first) -- the generator the SURVEY.md
    8(c) probe used for makeRandomVector()."""
    rs = np.random.RandomState(seed)
    raw = rs._bit_generator.random_raw(2 * 3 * n).astype(np.float64)
    lo, hi = raw[0::2], raw[1::2]
    can = (lo + hi * 4294967296.0) / 18446744073709551616.0
    v = (can * 2.0 + (-1.0)).reshape(n, 3)
    nrm = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])
    return v / nrm[:, None]


def drive(planner, scene, n_ticks, dynamic=False, advance=None, until_reached=False):
    """Run the planCallback sequence n_ticks times. Returns per-tick best
    indices and real-agent positions."""
    obs = scene["obstacles"].copy()
    best, pos = [], []
    for t in range(n_ticks):
        b = planner.tick(obs, scene["dt"], scene["cost_gains"], scene["ws_limits"])
        if dynamic:
            obs = advance(obs)
        best.append(np.asarray(b).copy())
        pos.append(np.asarray(planner.real_state()[0]).copy())
        if until_reached and np.all(np.asarray(planner.dist_from_goal()) < 0.01):
            break
    return np.asarray(best), np.asarray(pos)

"""The road from "parity unpinned" to "pinned" (VERDICT r4 item 3).

oracle/pin/make_pin.sh compiles the reference's own cf_agent.cpp + cf_manager.cpp (by path, unmodified) with the
build-owned driver oracle/pin/pin_harness.cpp against a REAL Eigen3 and REAL dqrobotics and writes
tests/golden/ref_<scenario>.json. Neither library exists in this repo's build container -- and a stand-in Eigen pins
nothing -- so here the script must SKIP (exit 77) without writing anything; on a machine that can build the reference
it produces the fixtures, and from then on test_oracle_is_held_to_the_reference_fixtures compares the oracle with them
bit for bit under BOTH dot-product associations and reports which one the reference build evaluates."""
import glob
import json
import os
import subprocess
import sys

import pytest

import conftest

ROOT = conftest.ROOT
PIN = os.path.join(ROOT, "oracle", "pin")
REFS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.json")))


def test_scenarios_are_current_and_parse():
    """oracle/pin/scenarios/*.txt are what make_scenarios.py writes from scenes.py, and replay.py reads them back"""
    sys.path.insert(0, PIN)
    import replay
    names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(PIN, "scenarios", "*.txt")))
    assert {"static1_n10_h100", "c1_static1_n16_h100", "static1_shipped", "dyn1_shipped", "trap_shipped", "dyn1_h300_hysteresis",
            "c2_64x200x32", "static1_closed_loop_lag30", "dyn1_two_goals", "dyn1_freq2"} <= set(names)
    before = {n: open(os.path.join(PIN, "scenarios", n + ".txt")).read() for n in names}
    subprocess.run([sys.executable, os.path.join(PIN, "make_scenarios.py")], check=True, capture_output=True)
    for n in names:
        assert open(os.path.join(PIN, "scenarios", n + ".txt")).read() == before[n], n + ": regenerate and commit"
        s = replay.load_scenario(os.path.join(PIN, "scenarios", n + ".txt"))
        per_init = max(0, s["n_agents"] - 5) * s["obstacles"].shape[0]
        assert len(s["random"]) == per_init * len(s["goals"]) and s["name"] == n
    s = replay.load_scenario(os.path.join(PIN, "scenarios", "trap_shipped.txt"))
    assert s["obstacles"].shape == (22, 7) and s["max_prediction_steps"] == 1500


def test_pin_recipe_skips_without_real_eigen_and_dqrobotics_or_writes_fixtures():
    """one command, exit 77 = skipped with the reason; never a stand-in build"""
    before = set(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.json")))
    r = subprocess.run(["bash", os.path.join(PIN, "make_pin.sh")] + (["/root/reference"] if os.path.isdir("/root/reference") else []),
                       capture_output=True, text=True, timeout=3600)
    if r.returncode == 77:
        assert "SKIPPED" in r.stderr and ("Eigen3" in r.stderr or "dqrobotics" in r.stderr or "reference checkout" in r.stderr), r.stderr
        assert set(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.json"))) == before
        assert not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "pin_harness"))
    else:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert len(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.json"))) >= 10


@pytest.mark.skipif(not REFS, reason="no tests/golden/ref_*.json yet: run `bash oracle/pin/make_pin.sh <reference checkout>` on a machine "
                                     "with Eigen3 + dqrobotics (cannot be produced in this image; PARITY UNPINNED until then)")
def test_oracle_is_held_to_the_reference_fixtures():
    matched = {}
    for ref in REFS:
        name = os.path.basename(ref)[4:-5]
        scn = os.path.join(PIN, "scenarios", name + ".txt")
        assert os.path.exists(scn), "fixture without its scenario: " + name
        meta = json.load(open(ref))["meta"]
        res = {}
        for variant in ("", "rassoc"):
            r = subprocess.run([sys.executable, os.path.join(PIN, "replay.py"), scn, ref], env=dict(os.environ, PMAF_VARIANT=variant),
                               capture_output=True, text=True, timeout=3600)
            assert r.returncode == 0, r.stderr[-3000:]
            res[variant] = json.loads(r.stdout.strip().splitlines()[-1])
        ok = [v for v in res if res[v]["match"]]
        print("%-28s Eigen %s vectorize=%s glibc %s cpu_fma=%s | left-assoc: %s  right-assoc: %s" % (
            name, meta["eigen"], meta["eigen_vectorize"], meta.get("glibc"), meta.get("cpu_fma"),
            "MATCH" if res[""]["match"] else "%d of %d differ (max %.3g, first %s)" % (res[""]["mismatches"], res[""]["compared"], res[""]["max_abs_diff"], res[""]["first"]),
            "MATCH" if res["rassoc"]["match"] else "%d of %d differ (max %.3g)" % (res["rassoc"]["mismatches"], res["rassoc"]["compared"], res["rassoc"]["max_abs_diff"])))
        assert ok, "%s: the oracle reproduces the reference under NEITHER association: %s" % (name, res)
        matched[name] = ok
    common = set.intersection(*[set(v) for v in matched.values()])
    assert common, "no single association reproduces every fixture: %s" % matched
    print("the reference build evaluates the %s association -> use %s" % (
        "RIGHT (a0 b0 + (a1 b1 + a2 b2))" if "rassoc" in common and "" not in common else "LEFT ((a0 b0 + a1 b1) + a2 b2)",
        "lib_rassoc/libpmaf_hip.so (PMAF_VARIANT=rassoc)" if "rassoc" in common and "" not in common else "lib/libpmaf_hip.so (the default)"))


class LockStep:
    """drives the HIP planner and the oracle (portable-exp mode: the kernels' own exp) through the same calls and requires
    every value they hand back to be bit-identical; returns the HIP planner's"""

    def __init__(self, hip, ora):
        self.hip, self.ora = hip, ora

    @staticmethod
    def _same(a, b):
        if isinstance(a, (tuple, list)):
            return len(a) == len(b) and all(LockStep._same(x, y) for x, y in zip(a, b))
        if a is None or b is None:
            return a is None and b is None
        import numpy as np
        a, b = np.asarray(a), np.asarray(b)
        return a.shape == b.shape and bool(np.all((a == b) | ((a != a) & (b != b))))

    def __getattr__(self, name):
        fh, fo = getattr(self.hip, name), getattr(self.ora, name)

        def call(*args, **kw):
            rh, ro = fh(*args, **kw), fo(*args, **kw)
            assert LockStep._same(rh, ro), "HIP and oracle differ in %s()" % name
            return rh
        return call


@pytest.mark.gpu
@pytest.mark.skipif(not REFS, reason="no tests/golden/ref_*.json yet (see test_oracle_is_held_to_the_reference_fixtures)")
def test_hip_path_is_held_to_the_reference_fixtures(pmaf, oracle):
    """The north star itself, once the fixtures exist -- the HIP planner (through the C-ABI) against the REFERENCE's own
    record, for the library variant whose evaluation order is the reference build's (the oracle of that order must
    reproduce the fixture bit for bit, else this variant is skipped). Per scenario:
      1. HIP == the oracle in its portable-exp mode, bit for bit, on every call (LockStep);
      2. both against the reference's record: same best-index sequence, set-points and the selected agent's scored
         trajectory within 1e-5 m. The one operation that is not a correctly rounded IEEE one is exp: the kernels restate
         glibc >= 2.28's FMA exp bit for bit (round 5), so against a fixture made on such a machine (any x86-64 glibc >= 2.28
         with FMA) the deviation printed below is 0. A fixture made with ANOTHER libm differs in exp's last bit now and
         then; scenarios where that alone is amplified past 1e-5 m -- the flow check with the round-4 exp found the second
         leg of dyn1_two_goals and the lagged closed loop on static1 -- are reported, not failed: a property of the scene
         and of the two libms, which no implementation escapes."""
    sys.path.insert(0, PIN)
    import replay
    order = pmaf.load_library().pmaf_eval_order()
    sensitive = []
    for ref_path in REFS:
        name = os.path.basename(ref_path)[4:-5]
        scn = replay.load_scenario(os.path.join(PIN, "scenarios", name + ".txt"))
        ref = json.load(open(ref_path))
        exact = replay.replay(scn, ref)                      # the oracle of THIS variant with libm exp, bit for bit
        if not exact["match"]:
            pytest.skip("evaluation order %d is not the reference build's (oracle differs on %s: %s) -- run with the other PMAF_VARIANT"
                        % (order, name, exact["first"]))
        oracle.set_exp_mode(1)
        try:
            res = replay.replay(scn, ref, tol=1e-5, selected_only=True, order=order,
                                make=lambda sc, ip: LockStep(pmaf.PmafPlanner(sc, device=0, mgr_init_pos=ip),
                                                             oracle.OraclePlanner(sc, mgr_init_pos=ip)))
        finally:
            oracle.set_exp_mode(0)
        print("%-28s HIP == oracle (portable exp) on every call; vs the reference: %d values over %d ticks, %d beyond 1e-5 m (max %.3g)%s" % (
            name, res["compared"], res["ticks"], res["mismatches"], res["max_abs_diff"],
            ("  (exact)" if res["max_abs_diff"] == 0 else "") if res["match"] else "  <- another libm's exp, amplified by this scene: " + str(res["first"])))
        if conftest.libm_is_restated(oracle):
            # the oracle with THIS host's libm reproduced the fixture bit for bit (above), this host's libm is the algorithm
            # the kernels restate, and HIP == the oracle call by call: nothing is left to tolerate
            assert res["match"] and res["max_abs_diff"] == 0, (name, res["first"])
        if not res["match"]:
            sensitive.append(name)
    assert len(sensitive) < len(REFS), "every scenario deviates from the reference: not an exp effect"
    # the well-conditioned core must hold: BASELINE C1 / C2, the shipped static1 / dyn1 / trap tasks
    assert not {"c1_static1_n16_h100", "c2_64x200x32", "static1_shipped", "dyn1_shipped", "trap_shipped"} & set(sensitive), sensitive

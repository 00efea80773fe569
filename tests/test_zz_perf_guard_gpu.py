"""Coarse performance guards (MI355X; the file sorts last so that under `pytest -x` a timing failure cannot hide a functional one): rollout-kernel time per launch, from HIP events on the kernel's own dispatch
(pmaf_set_profiling -- the device's clock, independent of the box's launch latency), against bounds ~20 % above the
round-5 records (profiles/r5_bench_*.json, profiles/r5_regime.txt). Not a benchmark: they exist because round 5 lost 45 %
on one kernel family to an LDS-occupancy cliff that no parity test could see (NOTES.md) -- a launch that drops a block
per CU, spills to scratch or falls to the generic kernel trips these."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def kernel_us(pmaf, scs, ticks=24, warm=6, bound=None):
    """best of up to three measurements (stops at the first one within `bound`): a guard must not fail on a busy box"""
    best = None
    for _ in range(3):
        us, cfg = _kernel_us(pmaf, scs, ticks, warm)
        best = us if best is None else min(best, us)
        if bound is None or best <= bound:
            break
    return best, cfg


def _kernel_us(pmaf, scs, ticks, warm):
    one = not isinstance(scs, list)
    sc = scs if one else scs[0]
    starts = sc["start"] if one else np.stack([s["start"] for s in scs])
    h = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    h.set_initial_position(starts)
    h.set_profiling(True)
    for _ in range(warm):
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop()
    h.reset_kernel_stats()
    for k in range(ticks):
        if k % 8 == 0:
            h.set_initial_position(starts)      # full-horizon rollouts only
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop()
    ms, n, steps = h.kernel_stats()
    cfg = h.launch_config()
    h.close()
    return ms / n * 1e3, cfg


# (name, record of round 5 in us, bound in us)
BASELINE_CASES = [("C1", 109.0, 131.0), ("C2", 224.5, 270.0), ("C3", 956.0, 1150.0), ("C4", 276.0, 331.0)]


@pytest.mark.parametrize("cfg,record,bound", BASELINE_CASES)
def test_baseline_config_kernel_time(pmaf, scenes, cfg, record, bound):
    scs = scenes.dual_arm_scenes() if cfg == "C4" else scenes.config_scene(cfg)
    us, lc = kernel_us(pmaf, scs, bound=bound)
    print("%s: %.1f us per rollout launch (round-5 record %.1f, bound %.1f), %r" % (cfg, us, record, bound, lc))
    assert us <= bound


def test_c5_eight_populations_kernel_time(pmaf, scenes):
    us, lc = kernel_us(pmaf, [scenes.config_scene("C5", scene_id=s) for s in range(8)], ticks=12, warm=3, bound=850.0)
    print("C5 x 8: %.1f us per rollout launch (round-5 record 710, bound 850), %r" % (us, lc))
    assert us <= 850.0


@pytest.mark.parametrize("n,m,record,bound", [(2048, 128, 640.0, 770.0), (4096, 128, 1185.0, 1420.0), (8192, 32, 700.0, 840.0),
                                               (1024, 128, 409.0, 490.0)])
def test_many_agent_kernel_time(pmaf, scenes, n, m, record, bound):
    """the agent-count sweep of tools/regime.py at its corners: one-wave two-slot kernel with every SIMD holding two waves
    (2048 x 128), its second round (4096 x 128), the group kernel (8192 x 32), the split kernel at one block per CU
    (1024 x 128); 200 steps, far goal"""
    us, lc = kernel_us(pmaf, scenes.synthetic_scene(n, 200, m, 2, 0), ticks=10, warm=3, bound=bound)
    print("%d agents x 200 steps x %d obstacles: %.1f us per launch (round-5 record %.1f, bound %.1f), %r" % (n, m, us, record, bound, lc))
    assert us <= bound

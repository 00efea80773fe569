"""Tolerance parity (north star: the SELECTED trajectory within 1e-5 m of the reference CPU planner) where bit
equality cannot hold by construction, over >= 50 closed-loop ticks at the BASELINE sizes:

  * the default (strict) kernels against the oracle in its reference-faithful mode 0 -- the platform libm's exp, as
    B/src/cf_agent.cpp:220 calls it; exp is the ONLY operation of the path that is not a correctly rounded IEEE one, and
    the kernels restate glibc >= 2.28's (round 5): on a host with that libm the comparison is asserted EXACT, elsewhere
    within 1e-5 m. tests/test_parity_gpu.py holds the same comparison for C1 / C2 (30 ticks) and the nine
    shipped task scenes (closed loop until reached); here: C3 (500-step chains through 128 obstacles -- where the
    survey measured chaotic amplification), C4 and C5 at full size.
  * the opt-in CONTRACTED policy (PMAF_FLAG_CONTRACTED: reciprocal / reciprocal-square-root sequences + FMA
    contraction, include/pmaf.h) against the same oracle mode, C1-C5 and the nine shipped task scenes.

Per run the two planners are ticked in lock step (same live obstacles); every tick compares, per population,
  - the best index (a difference is a FLIP: recorded with the oracle's cost of both candidates, and the population is
    dropped from the later comparisons -- the two closed loops are then different experiments),
  - the set-point (the real agent's position after its step),
  - the selected trajectory = the predicted path of the agent the tick selected, as it was scored,
  - how many of the NON-selected agents' paths differ by more than 1e-5 m (reported, not asserted: a chaotic
    rollout amplifies any last-bit perturbation, a different libm included).
The summary line of every run is printed (pytest -s) and appended to $PMAF_TOL_REPORT when that is set."""
import json
import os

import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu

TOL = 1e-5  # metres, BASELINE.json north_star


def _oracle_threads():
    return max(1, min(32, (os.cpu_count() or 2) // 2))


def _paths4(planner_paths, P):
    p, n = planner_paths
    p = np.asarray(p)
    n = np.asarray(n)
    if p.ndim == 3:
        p, n = p[None], n[None]
    assert p.shape[0] == P
    return p, n


def lockstep(hip, oras, scs, ticks, live_fn=None, until_reached=False):
    """Tick `hip` (P populations) and the P oracles in lock step. live_fn(t, pos_hip, pos_ora) -> (obs_hip, obs_ora)
    supplies the live obstacle lists ([P][n_obs][7] each); default: the scenes' own lists, unchanged."""
    P = len(oras)
    sc = scs[0]
    thr = _oracle_threads()
    st = dict(ticks=0, flips=[], max_setpoint=0.0, max_selected=0.0, max_nonselected=0.0, nonselected_over=0,
              nonselected_compared=0, length_mismatch=0)
    alive = [True] * P
    obs0 = np.stack([s["obstacles"] for s in scs])
    pos_h = np.stack([s["start"] for s in scs]).astype(np.float64)
    pos_o = pos_h.copy()
    for t in range(ticks):
        hip.stop()
        ph, nh = _paths4(hip.paths(), P)            # the paths this tick's selection scores
        oh, oo = live_fn(t, pos_h, pos_o) if live_fn else (obs0, obs0)
        bh = np.atleast_1d(hip.tick(oh if P > 1 else oh[0], sc["dt"], sc["cost_gains"], sc["ws_limits"]))
        rh = np.asarray(hip.real_state()[0]).reshape(P, 3)
        for p, o in enumerate(oras):
            if not alive[p]:
                continue
            po, no = o.paths()
            bo = o.tick_omp(oo[p], sc["dt"], sc["cost_gains"], sc["ws_limits"], thr) if thr > 1 else \
                o.tick(oo[p], sc["dt"], sc["cost_gains"], sc["ws_limits"])
            ro = np.asarray(o.real_state()[0])
            if int(bh[p]) != int(bo):
                co = o.costs()
                st["flips"].append(dict(tick=t, pop=p, hip=int(bh[p]), oracle=int(bo), oracle_cost_of_hip_choice=float(co[bh[p]]),
                                        oracle_cost_of_its_choice=float(co[bo]),
                                        rel_margin=float(abs(co[bh[p]] - co[bo]) / max(abs(co[bo]), 1e-300))))
                alive[p] = False
                continue
            pos_o[p] = ro
            st["max_setpoint"] = max(st["max_setpoint"], float(np.abs(rh[p] - ro).max()))
            k = min(int(nh[p, bo]), int(no[bo]))
            if nh[p, bo] != no[bo]:
                st["length_mismatch"] += 1
            st["max_selected"] = max(st["max_selected"], float(np.abs(ph[p, bo, :k] - po[bo, :k]).max()))
            kk = np.minimum(nh[p], no)
            idx = np.arange(ph.shape[2])[None, :] < kk[:, None]
            dev = np.where(idx[:, :, None], np.abs(ph[p] - po), 0.0).max(axis=(1, 2))
            dev[bo] = 0.0
            st["max_nonselected"] = max(st["max_nonselected"], float(dev.max()))
            st["nonselected_over"] += int((dev > TOL).sum())
            st["nonselected_compared"] += len(dev) - 1
        pos_h = rh.copy()
        st["ticks"] = t + 1
        if not any(alive):
            break
        if until_reached and all(d < 0.01 for d in np.atleast_1d(hip.dist_from_goal())):
            break
    st["populations_compared_to_the_end"] = int(sum(alive))
    return st


def report(name, policy, st):
    line = dict(case=name, policy=policy, **st)
    print("%-26s %-10s %3d ticks | flips %d | set-point %.3g m | selected trajectory %.3g m | non-selected > 1e-5 m: %d of %d (max %.3g m)"
          % (name, policy, st["ticks"], len(st["flips"]), st["max_setpoint"], st["max_selected"], st["nonselected_over"],
             st["nonselected_compared"], st["max_nonselected"]))
    for f in st["flips"]:
        print("     flip:", f)
    out = os.environ.get("PMAF_TOL_REPORT")
    if out:
        with open(out, "a") as fh:
            fh.write(json.dumps(line) + "\n")


def check(st, north_star=True):
    """north_star: the contract -- selected trajectory and set-point within 1e-5 m over the whole run; a flip is tolerated
    only as a tie (the two candidates' costs within 1e-9 relative in the oracle's own evaluation), and is listed either
    way. north_star=False (the cases where the CONTRACTED policy was measured NOT to meet the bar -- chaotic scenes,
    see CONTRACTED_EXCEEDS): only what still holds is asserted -- while the two planners select the same agent the
    published set-points agree (the real agent's step is always evaluated in strict arithmetic); the deviations are
    reported."""
    assert st["max_setpoint"] <= TOL
    if not north_star:
        return
    for f in st["flips"]:
        assert f["rel_margin"] <= 1e-9, "best-index flip with a real cost margin: %r" % (f,)
    assert st["max_selected"] <= TOL
    assert st["length_mismatch"] == 0


def _build(pmaf, oracle, scs, **kw):
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs if len(scs) > 1 else scs[0], device=0, mgr_init_pos=starts if len(scs) > 1 else starts[0], **kw)
    hip.set_initial_position(starts if len(scs) > 1 else starts[0])
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    return hip, oras


def _config(pmaf, scenes, cfg):
    """-> (scenes of the populations, live_fn or None)"""
    if cfg == "C4":
        scs = scenes.dual_arm_scenes()
        obs = np.stack([s["obstacles"] for s in scs])
        ch, co = pmaf.shard.DualArmCoupling(obs, 0.1), pmaf.shard.DualArmCoupling(obs, 0.1)
        return scs, (lambda t, ph, po: (ch.coupled_obstacles(ph), co.coupled_obstacles(po)))
    if cfg == "C5":
        return [scenes.config_scene("C5", scene_id=s) for s in range(8)], None
    return [scenes.config_scene(cfg)], None


POLICIES = {"strict": {}, "contracted": {"contracted": True}}

# Where the contracted policy was MEASURED not to meet the north star (round 4, profiles/r4_tolerance_report.jsonl;
# deterministic: same inputs, same kernels, same bits on every box). All of them are scenes whose rollouts are chaotic --
# the strict kernels against another libm's exp already show divergent NON-selected agents there -- and the contracted
# arithmetic perturbs ~30x more operations per step:
#   C5 (scene 1 of the 8): selected trajectory 3.2e-3 m, 10 % of that population's rollouts differ by > 1e-5 m
#   sim_kobo_dyn_spheres1 / 2 (H = 1500, moving spheres): selected trajectory 0.22 / 0.23 m, no best-index difference,
#     set-point sequence identical over 900 ticks
#   sim_kobo_dyn_spheres3 (H = 1200): a best-index difference at tick 1 -- where the ORACLE's own evaluation-order
#     variants flip too (tools/oracle_conditioning.py, profiles/r4_oracle_conditioning.txt)
# The policy is opt-in for exactly this reason; include/pmaf.h and DESIGN.md say where it holds.
CONTRACTED_EXCEEDS = {"C5", "sim_kobo_dyn_spheres1", "sim_kobo_dyn_spheres2", "sim_kobo_dyn_spheres3"}


@pytest.mark.parametrize("cfg,ticks", [("C3", 60), ("C4", 120), ("C5", 50)])
def test_strict_kernels_against_libm_oracle_full_size(pmaf, oracle, scenes, cfg, ticks, request):
    """the reference-faithful comparison (std::exp, B/src/cf_agent.cpp:220) where it can fail: long chains (C3), the
    repulsive obstacle in range (C4: 120 ticks, the arms pass each other), the group kernel at full size (C5)"""
    conftest.expect_chaotic(request, "strict_libm:" + cfg)
    oracle.set_exp_mode(0)
    scs, live = _config(pmaf, scenes, cfg)
    hip, oras = _build(pmaf, oracle, scs)
    st = lockstep(hip, oras, scs, ticks, live)
    report(cfg, "strict", st)
    check(st)
    if conftest.libm_is_restated(oracle):
        # round 5: the kernels' exp is glibc's, bit for bit -- on such a host nothing is left to tolerate: every agent of
        # every population, selected or not, equals the reference-faithful oracle exactly
        assert not st["flips"] and st["max_setpoint"] == 0.0 and st["max_selected"] == 0.0 and st["max_nonselected"] == 0.0
    hip.close()


@pytest.mark.parametrize("cfg,ticks", [("C1", 60), ("C2", 60), ("C3", 60), ("C4", 120), ("C5", 50)])
def test_contracted_policy_within_north_star_tolerance(pmaf, oracle, scenes, cfg, ticks, request):
    conftest.expect_chaotic(request, "contracted:" + cfg)
    oracle.set_exp_mode(0)
    scs, live = _config(pmaf, scenes, cfg)
    hip, oras = _build(pmaf, oracle, scs, contracted=True)
    st = lockstep(hip, oras, scs, ticks, live)
    report(cfg, "contracted", st)
    check(st, north_star=cfg not in CONTRACTED_EXCEEDS)
    hip.close()


def test_contracted_policy_on_the_sliced_wave_per_agent_kernel(pmaf, oracle, scenes):
    """k_rollout_w64_sliced<3>: the contracted arithmetic policy with two waves per SIMD (BASELINE C5's per-GPU load at 4 GPUs,
    scenes 0 and 2 -- scene 1 is the chaotic one, CONTRACTED_EXCEEDS). The slices are scheduling only: the contracted results of
    this launch must equal, bit for bit, those of the SAME populations run one per handle (1 024 waves, one per SIMD, no
    slicing), and stay within the policy's tolerance contract against the libm oracle."""
    oracle.set_exp_mode(0)
    scs = [scenes.config_scene("C5", scene_id=s) for s in (0, 2)]
    hip, oras = _build(pmaf, oracle, scs, contracted=True)
    lc = hip.launch_config()
    assert (lc["lanes_per_agent"], lc["priority_slices"]) == (64, True), lc
    singles = []
    for q in scs:
        h1, _ = _build(pmaf, oracle, [q], contracted=True)
        assert not h1.launch_config()["priority_slices"]
        singles.append(h1)
    sc = scs[0]
    obs = np.stack([q["obstacles"] for q in scs])
    for t in range(5):
        bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        for i, h1 in enumerate(singles):
            assert h1.tick(obs[i], sc["dt"], sc["cost_gains"], sc["ws_limits"]) == bh[i]
    hip.stop()
    ph, nh = hip.paths()
    for i, h1 in enumerate(singles):
        h1.stop()
        p1, n1 = h1.paths()
        np.testing.assert_array_equal(nh[i], n1)
        np.testing.assert_array_equal(ph[i], p1)
        np.testing.assert_array_equal(hip.costs()[i], h1.costs())
        h1.close()
    hip.close()
    hip, oras = _build(pmaf, oracle, scs, contracted=True)
    st = lockstep(hip, oras, scs, 20, None)
    report("C5x2", "contracted", st)
    check(st)
    hip.close()


def _task_records():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "task_scenes.json")))


@pytest.mark.parametrize("task", sorted(_task_records()))
def test_contracted_policy_on_the_shipped_task_scenes(pmaf, oracle, scenes, task, request):
    """the reference's own operating point (10 agents, H = 1500 / 1200, moving obstacles), closed loop until reached
    or 900 ticks"""
    conftest.expect_chaotic(request, "contracted_task:" + task)
    oracle.set_exp_mode(0)
    sc = scenes.scene_from_record(_task_records()[task], task)
    hip, oras = _build(pmaf, oracle, [sc], contracted=True)
    state = {"obs": sc["obstacles"].copy()}

    def live(t, ph, po):
        o = state["obs"]
        state["obs"] = scenes.advance_live_obstacles(o)
        return o[None], o[None]

    st = lockstep(hip, oras, [sc], 900, live, until_reached=True)
    report(task, "contracted", st)
    check(st, north_star=task not in CONTRACTED_EXCEEDS)
    hip.close()

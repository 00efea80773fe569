"""Parity of the HIP path (through the C-ABI, libpmaf_hip.so) against the CPU
oracle on identical seeded inputs. Bar (BASELINE.json north_star): selected
trajectory within 1e-5 m of the reference CPU planner.

Two oracle modes (oracle/pmaf_oracle.c):
  * portable exp (mode 1): the oracle evaluates exp() with the same
    table-free algorithm as the kernels. Every other operation is a correctly
    rounded IEEE + - * / sqrt in the same order on both sides, so the HIP
    results must be BIT-IDENTICAL: TOL = 0 on every trajectory point, cost,
    rotation vector and flag, at every horizon. All tests below run in this
    mode unless they say otherwise.
  * libm exp (mode 0, the reference-faithful restatement): glibc's exp differs
    from any other exp in the last bit on ~5-10 % of arguments, and the
    rollout amplifies a 1-ulp perturbation (mildly over the BASELINE horizons,
    chaotically over 1000+ steps). test_libm_exp_oracle_* assert the north-star
    tolerance 1e-5 m on the BASELINE configs in that mode."""
import copy
import os

import numpy as np
import pytest

EXP_OVERFLOW = float.fromhex("0x1.62e42fefa39efp+9")   # exp()'s overflow threshold, 709.782712893384

import conftest
from conftest import drive

pytestmark = pytest.mark.gpu

TOL = 0.0        # metres, portable-exp oracle: bit-identical
LIBM_TOL = 1e-5  # metres, libm-exp oracle: the north-star tolerance


@pytest.fixture(autouse=True)
def _portable_exp_oracle(oracle):
    oracle.set_exp_mode(1)
    yield
    oracle.set_exp_mode(0)


def make_pair(pmaf, oracle, scene, **kw):
    hip = pmaf.PmafPlanner(scene, device=0, mgr_init_pos=scene["start"], **kw)
    ora = oracle.OraclePlanner(scene, mgr_init_pos=scene["start"])
    hip.set_initial_position(scene["start"])
    ora.set_initial_position(scene["start"])
    return hip, ora


def assert_state_equal(hip, ora, tol=TOL):
    ph, nh = hip.paths()
    po, no = ora.paths()
    np.testing.assert_array_equal(nh, no)
    assert np.array_equal(np.isnan(ph), np.isnan(po))
    m = ~np.isnan(po)
    assert np.abs(ph[m] - po[m]).max() <= tol
    np.testing.assert_allclose(hip.min_obs_dist(), ora.min_obs_dist(), rtol=0, atol=tol)
    np.testing.assert_allclose(hip.agent_vel(), ora.agent_vel(), rtol=0, atol=tol)
    np.testing.assert_array_equal(hip.success(), ora.success())
    np.testing.assert_allclose(hip.path_lengths(), ora.path_lengths(), rtol=0, atol=tol)
    np.testing.assert_array_equal(hip.known(), ora.known())
    np.testing.assert_allclose(hip.rot_vecs(), ora.rot_vecs(), rtol=0, atol=tol, equal_nan=True)
    rh, ro = hip.real_state(), ora.real_state()
    for a, b in zip(rh, ro):
        np.testing.assert_allclose(a, b, rtol=0, atol=tol, equal_nan=True)
    kh, rrh = hip.real_known()
    ko, rro = ora.real_known()
    np.testing.assert_array_equal(kh, ko)
    np.testing.assert_allclose(rrh, rro, rtol=0, atol=tol, equal_nan=True)


def run_both(pmaf, oracle, scenes, scene, n_ticks, dynamic=False, **kw):
    hip, ora = make_pair(pmaf, oracle, scene, **kw)
    bh, ph = drive(hip, scene, n_ticks, dynamic, scenes.advance_live_obstacles)
    bo, po = drive(ora, scene, n_ticks, dynamic, scenes.advance_live_obstacles)
    hip.stop()
    np.testing.assert_array_equal(bh, bo)
    assert np.abs(ph - po).max() <= TOL
    np.testing.assert_allclose(hip.costs(), ora.costs(), rtol=0, atol=0)
    assert_state_equal(hip, ora)
    assert hip.best_type() == ora.best_type() and hip.best_id() == ora.best_id()
    return hip, ora


def test_c1_static1_16_agents(pmaf, oracle, scenes):
    """BASELINE config C1: static1 scene, 16 agents, 100-step horizon."""
    sc = scenes.config_scene("C1")
    hip, ora = run_both(pmaf, oracle, scenes, sc, 25)
    np.testing.assert_allclose(hip.real_path(), ora.real_path(), rtol=0, atol=TOL)
    hip.close()


def test_static1_as_shipped_10_agents_long_horizon(pmaf, oracle, scenes):
    """the task file as shipped: 10 agents, max_prediction_steps 1500 ->
    agents stop early at distGoal <= 0.1 (cf_agent.cpp:310)"""
    sc = scenes.static1_scene(10, 1499)
    hip, ora = run_both(pmaf, oracle, scenes, sc, 6)
    assert (hip.n_points() < 1500).any()
    hip.close()


@pytest.mark.parametrize("lpa", [1, 2, 4, 8, 16, 32, 64])
def test_every_lane_mapping(pmaf, oracle, scenes, lpa):
    """all lanes-per-agent instantiations give the same answer; N=13 leaves a
    ragged last wave, M=9 is not a multiple of any LPA > 1"""
    sc = scenes.static1_scene(13, 120)
    hip, _ = run_both(pmaf, oracle, scenes, sc, 8, lanes_per_agent=lpa)
    assert hip.launch_config()["lanes_per_agent"] == lpa
    hip.close()


def test_c2_synthetic_64_agents_32_obstacles(pmaf, oracle, scenes):
    sc = scenes.config_scene("C2")
    hip, _ = run_both(pmaf, oracle, scenes, sc, 12)
    hip.close()


def test_c2_dynamic_obstacles(pmaf, oracle, scenes):
    """moving obstacles exercise predictObstacles (cf_agent.cpp:270-276) and
    the live-obstacle stream (dynamic_obstacle_node.cpp:355-357)"""
    sc = scenes.config_scene("C2", scene_id=3, dynamic=True)
    hip, _ = run_both(pmaf, oracle, scenes, sc, 12, dynamic=True)
    hip.close()


@pytest.mark.parametrize("one_wave", [False, True])
@pytest.mark.parametrize("lpa", [0, 8, 64])
def test_c3_256_agents_128_obstacles(pmaf, oracle, scenes, lpa, monkeypatch, one_wave):
    """BASELINE config C3 (LDS-tiled obstacle sweep: several tiles per lane)"""
    if one_wave:   # the one-wave kernels (2 / 4 obstacle slots per lane) instead of k_rollout_mw
        monkeypatch.setenv("PMAF_MW", "0")
    sc = scenes.config_scene("C3")
    hip, _ = run_both(pmaf, oracle, scenes, sc, 2, lanes_per_agent=lpa)
    hip.close()


def test_dyn1_closed_loop_until_reached(pmaf, oracle, scenes):
    """dual_arms_dyn1 scene driven until getDistFromGoal() < 0.01 (taskCallback
    'reached', panda_bimanual_control.cpp:565-569): ~745 ticks with moving
    obstacles, early termination and hysteresis."""
    sc = scenes.dyn1_scene(10, 1500)
    hip, ora = make_pair(pmaf, oracle, sc)
    bh, ph = drive(hip, sc, 2000, True, scenes.advance_live_obstacles, until_reached=True)
    bo, po = drive(ora, sc, 2000, True, scenes.advance_live_obstacles, until_reached=True)
    assert len(bh) == len(bo)
    np.testing.assert_array_equal(bh, bo)
    assert np.abs(ph - po).max() <= TOL
    hip.close()


def test_batched_populations_match_individual_oracles(pmaf, oracle, scenes):
    """P independent populations in one handle (BASELINE C5 shape, reduced N)"""
    scs = [scenes.synthetic_scene(40, 150, 32, 5, sid, dynamic=(sid % 2 == 1)) for sid in range(5)]
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=np.stack([s["start"] for s in scs]))
    hip.set_initial_position(np.stack([s["start"] for s in scs]))
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    obs = np.stack([s["obstacles"] for s in scs])
    for t in range(6):
        bh = hip.tick(obs, scs[0]["dt"], scs[0]["cost_gains"], scs[0]["ws_limits"])
        bo = [o.tick(obs[i], scs[0]["dt"], scs[0]["cost_gains"], scs[0]["ws_limits"]) for i, o in enumerate(oras)]
        np.testing.assert_array_equal(bh, bo)
        obs = np.stack([scenes.advance_live_obstacles(o) if i % 2 == 1 else o for i, o in enumerate(obs)])
    ph, nh = hip.paths()
    for i, o in enumerate(oras):
        po, no = o.paths()
        np.testing.assert_array_equal(nh[i], no)
        assert np.abs(ph[i] - po).max() <= TOL
        np.testing.assert_allclose(hip.real_state()[0][i], o.real_state()[0], rtol=0, atol=TOL)
    hip.close()


def test_step_api_equals_fused_tick(pmaf, oracle, scenes):
    """stopPrediction / evaluateAgents / moveRealEEAgent / resetEEAgents /
    startPrediction called one by one == pmaf_tick == oracle"""
    sc = scenes.config_scene("C1")
    hip, ora = make_pair(pmaf, oracle, sc)
    for t in range(6):
        hip.stop()
        bh = hip.evaluate(sc["cost_gains"], sc["ws_limits"])
        hip.move_real(sc["obstacles"], sc["dt"], 1, bh)
        p, v, _ = hip.real_state()
        hip.reset_agents(p, v, sc["obstacles"])
        hip.start()
        bo = ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        assert bh == bo
    hip.stop()
    assert_state_equal(hip, ora)
    hip.close()


def test_rollout_before_any_evaluate_and_rescoring(pmaf, oracle, scenes):
    """start without a prior evaluate, then evaluate with two different
    workspace boxes (forces the k_score re-scoring path)"""
    sc = scenes.config_scene("C1")
    hip, ora = make_pair(pmaf, oracle, sc)
    hip.rollout()
    ora.rollout()
    assert_state_equal(hip, ora)
    tight = np.array([0.2, -0.2, 0.05, -0.05, 0.8, 0.6])
    for ws in (sc["ws_limits"], tight, sc["ws_limits"]):
        # evaluate mutates the hysteresis state identically on both sides
        assert hip.evaluate(sc["cost_gains"], ws) == ora.evaluate(sc["cost_gains"], ws)
        np.testing.assert_allclose(hip.costs(), ora.costs(), rtol=0, atol=0)
    hip.close()


def test_all_heuristic_types_population(pmaf, oracle, scenes):
    """explicit agent_types: 6 agents of each heuristic in a cluttered scene"""
    types = np.repeat([1, 2, 3, 4, 5, 6], 6).astype(np.int32)
    sc = scenes.synthetic_scene(36, 200, 20, 7, 1, agent_types=types)
    hip, _ = run_both(pmaf, oracle, scenes, sc, 5)
    hip.close()


def test_edge_only_sentinel_obstacle(pmaf, oracle, scenes):
    """M = 0: obstacle list holds only the trailing repulsive obstacle; every
    heuristic degenerates to the damped straight line"""
    sc = scenes.synthetic_scene(8, 80, 0, 9, 0)
    hip, ora = run_both(pmaf, oracle, scenes, sc, 3)
    ph, _ = hip.paths()
    assert np.abs(ph - ph[0]).max() == 0.0
    hip.close()


def test_edge_single_field_obstacle(pmaf, oracle, scenes):
    """one field obstacle: Obstacle/GoalObstacle 'closest other' is itself
    (cf_agent.cpp:434-446) -> zero vectors / NaNs must match the oracle"""
    sc = scenes.synthetic_scene(8, 120, 1, 9, 1)
    sc["obstacles"][0, :3] = [0.0, 0.02, 0.72]
    hip, _ = run_both(pmaf, oracle, scenes, sc, 3)
    hip.close()


def test_edge_had_degenerate_obstacle_on_goal_line(pmaf, oracle, scenes):
    """Had heuristic divides by |d x g| unguarded (cf_agent.cpp:609): obstacle
    centre exactly on the agent-goal line gives NaN; NaN pattern must match"""
    sc = scenes.synthetic_scene(6, 60, 1, 9, 2)
    sc["obstacles"][0] = [0.0, 0.0, 0.7, 0, 0, 0, 0.05]
    hip, ora = make_pair(pmaf, oracle, sc)
    for _ in range(2):
        hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    ph, nh = hip.paths()
    po, no = ora.paths()
    np.testing.assert_array_equal(nh, no)
    assert np.array_equal(np.isnan(ph), np.isnan(po))
    m = ~np.isnan(po)
    assert np.abs(ph[m] - po[m]).max() <= TOL
    hip.close()


def test_edge_start_inside_goal_region_and_capacity_one(pmaf, oracle, scenes):
    """guard false at once (distGoal <= 0.1) -> 1-point paths; capacity 1"""
    sc = scenes.synthetic_scene(6, 50, 4, 9, 3)
    sc["start"] = sc["goal"] + np.array([0.05, 0.0, 0.0])
    hip, _ = run_both(pmaf, oracle, scenes, sc, 3)
    assert (hip.n_points() == 1).all()
    hip.close()
    sc = scenes.synthetic_scene(6, 0, 4, 9, 4)  # max_prediction_steps = 1
    hip, _ = run_both(pmaf, oracle, scenes, sc, 3)
    hip.close()


def test_link_force_matches_oracle(pmaf, oracle, scenes):
    sc = scenes.config_scene("C1")
    sc["obstacles"][-1, :3] = [0.1, 0.0, 0.7]  # bring the repulsive obstacle into range
    hip, ora = make_pair(pmaf, oracle, sc)
    rng = np.random.default_rng(5)
    lp = rng.uniform(-0.3, 0.5, (37, 3)) + np.array([0.0, 0.0, 0.6])
    k = rng.uniform(0.01, 0.05, 37)
    np.testing.assert_allclose(hip.link_force(lp, k, sc["obstacles"]), ora.link_force(lp, k, sc["obstacles"]),
                               rtol=0, atol=0)
    hip.close()


def test_error_reporting(pmaf, scenes):
    sc = scenes.config_scene("C1")
    hip = pmaf.PmafPlanner(sc, device=0)
    with pytest.raises(pmaf.PmafError) as e:
        hip.move_real(sc["obstacles"], 0.01, 1, 0)  # no best agent yet
    assert e.value.code == -3
    with pytest.raises(pmaf.PmafError):
        pmaf.PmafPlanner(sc, device=0, lanes_per_agent=3)
    hip.close()


# ---------------------------------------------------------------------------
# reference-faithful oracle (libm exp): north-star tolerance on BASELINE configs
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("one_wave", [False, True])
@pytest.mark.parametrize("dynamic", [False, True])
def test_c3_twelve_ticks_latches_and_hysteresis(pmaf, oracle, scenes, dynamic, monkeypatch, one_wave):
    """C3 in depth: 12 ticks (12 rollouts of 256 agents x 500 steps through 128
    obstacles, 11 of them scored) so that rotation vectors latched in one tick
    persist into the next, known flags travel real agent -> agents, and the
    0.9 hysteresis is exercised at M = 128; static and moving obstacles"""
    if one_wave:   # the one-wave kernels (2 / 4 obstacle slots per lane) instead of k_rollout_mw
        monkeypatch.setenv("PMAF_MW", "0")
    sc = scenes.config_scene("C3", scene_id=1, dynamic=dynamic)
    hip, ora = run_both(pmaf, oracle, scenes, sc, 12, dynamic=dynamic)
    # latches did persist: some agents know obstacles the real agent does not know yet
    assert hip.known().sum() > 256 * hip.real_known()[0].sum()
    hip.close()


def test_c5_dynamic_obstacles_full_size(pmaf, oracle, scenes):
    """C5 at full size with MOVING obstacles streamed per tick (group kernel,
    2 waves per SIMD): 4 ticks, all 8192 agents' paths against 8 oracles"""
    scs = [scenes.config_scene("C5", scene_id=s, dynamic=True) for s in range(8)]
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    obs = np.stack([s["obstacles"] for s in scs])
    sc = scs[0]
    for t in range(4):
        bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = [o.tick(obs[i], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for i, o in enumerate(oras)]
        np.testing.assert_array_equal(bh, bo)
        np.testing.assert_array_equal(hip.real_state()[0], np.stack([o.real_state()[0] for o in oras]))
        obs = np.stack([scenes.advance_live_obstacles(o) for o in obs])
    ph, nh = hip.paths()
    rot = hip.rot_vecs()
    for i, o in enumerate(oras):
        po, no = o.paths()
        np.testing.assert_array_equal(nh[i], no)
        np.testing.assert_array_equal(ph[i], po)
        np.testing.assert_array_equal(hip.costs()[i], o.costs())
        np.testing.assert_array_equal(rot[i], o.rot_vecs())
    hip.close()


@pytest.mark.parametrize("dynamic,coupled_sentinel", [(False, False), (True, False), (True, True)])
def test_c5_two_scenes_per_gpu_run_the_sliced_wave_per_agent_kernel(pmaf, oracle, scenes, dynamic, coupled_sentinel):
    """BASELINE C5's per-GPU load at 4 GPUs: two scenes x 1024 agents in one handle = 2048 one-slot wave-per-agent rollouts,
    two per SIMD, which trade issue priority in time slices (k_rollout_w64_sliced: scheduling only). Six ticks at full size,
    every path point / cost / rotation vector of all 2048 agents against two oracles, tolerance 0 -- static obstacles (the
    STATIC-velocity loops), obstacles streamed per tick, and with the trailing repulsive obstacle flying through the scene
    (the loops that carry the lane-60 rider)."""
    scs = [copy.deepcopy(scenes.config_scene("C5", scene_id=s, dynamic=dynamic)) for s in (0, 1)]
    if coupled_sentinel:
        for q in scs:   # a repulsive obstacle that is in range of the rollouts: near the straight line, moving
            q["obstacles"][-1] = np.array([0.0, 0.05, 0.72, -0.03, 0.01, 0.0, 0.08])
    starts = np.stack([q["start"] for q in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    lc = hip.launch_config()
    assert (lc["lanes_per_agent"], lc["waves_per_agent"], lc["priority_slices"]) == (64, 1, True), lc
    oras = []
    for q in scs:
        o = oracle.OraclePlanner(q, mgr_init_pos=q["start"])
        o.set_initial_position(q["start"])
        oras.append(o)
    obs = np.stack([q["obstacles"] for q in scs])
    sc = scs[0]
    for t in range(6):
        bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = [o.tick(obs[i], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for i, o in enumerate(oras)]
        np.testing.assert_array_equal(bh, bo)
        np.testing.assert_array_equal(hip.real_state()[0], np.stack([o.real_state()[0] for o in oras]))
        if dynamic:
            obs = np.stack([scenes.advance_live_obstacles(o) for o in obs])
            if coupled_sentinel:
                obs[:, -1, 0:3] += obs[:, -1, 3:6] * sc["dt"]
    ph, nh = hip.paths()
    rot = hip.rot_vecs()
    for i, o in enumerate(oras):
        po, no = o.paths()
        np.testing.assert_array_equal(nh[i], no)
        np.testing.assert_array_equal(ph[i], po)
        np.testing.assert_array_equal(hip.costs()[i], o.costs())
        np.testing.assert_array_equal(rot[i], o.rot_vecs())
        np.testing.assert_array_equal(hip.min_obs_dist()[i], o.min_obs_dist())
    hip.close()


@pytest.mark.parametrize("cfg,ticks", [("C1", 30), ("C2", 30), ("C3", 2)])
def test_libm_exp_oracle_within_north_star_tolerance(pmaf, oracle, scenes, cfg, ticks, request):
    conftest.expect_chaotic(request, "libm:" + cfg)
    oracle.set_exp_mode(0)
    LIBM_TOL = conftest.libm_tol(oracle, 1e-5)      # 0 on a glibc >= 2.28 host: the kernels' exp is that libm's
    sc = scenes.config_scene(cfg)
    hip, ora = make_pair(pmaf, oracle, sc)
    bh, ph = drive(hip, sc, ticks)
    bo, po = drive(ora, sc, ticks)
    hip.stop()
    np.testing.assert_array_equal(bh, bo)
    # the selected trajectory = the real agent's path ...
    assert np.abs(ph - po).max() <= LIBM_TOL
    # ... and the winning agent's predicted path; report all agents too
    pth, nh = hip.paths()
    pto, no = ora.paths()
    np.testing.assert_array_equal(nh, no)
    best = int(bh[-1])
    assert np.abs(pth[best] - pto[best]).max() <= LIBM_TOL
    print("%s: max |path - libm oracle| best agent %.3g m, all agents %.3g m" %
          (cfg, np.abs(pth[best] - pto[best]).max(), np.abs(pth - pto).max()))
    assert np.abs(pth - pto).max() <= LIBM_TOL
    hip.close()


def test_device_arithmetic_is_ieee_exact(pmaf, oracle):
    """the parity argument: device / sqrt * + are correctly rounded (== host),
    device exp == the oracle's portable exp bit for bit"""
    rng = np.random.default_rng(1)
    n = 1_000_000
    a = rng.uniform(1e-6, 4.0, n)
    b = rng.uniform(1e-6, 4.0, n)
    assert (pmaf.debug_math(0, a, b) == a / b).all()
    assert (pmaf.debug_math(1, a) == np.sqrt(a)).all()
    assert (pmaf.debug_math(3, a, b) == a * b).all()
    assert (pmaf.debug_math(4, a, b) == a + b).all()
    x = np.concatenate([-rng.uniform(0.0, 3.0, 4_000_000), -rng.uniform(0.0, 600.0, 1_000_000), rng.uniform(0.0, 720.0, 500_000),
                        -np.ldexp(rng.uniform(0.5, 1.0, 500_000), -rng.integers(0, 70, 500_000)), [0.0, -0.0, -37.4, -500.0, -1e9, 709.7, 710.0],
                        # the top of the range (glibc's specialcase, x >= 512; (709.7800, 709.7827] was a NaN before round 6)
                        rng.uniform(511.0, 513.0, 50_000), rng.uniform(709.7, 709.79, 200_000),
                        [512.0, 709.781, 709.7827, EXP_OVERFLOW, np.nextafter(EXP_OVERFLOW, 1e9), 1023.0, 1024.0, 1e300, np.inf]])
    dev = pmaf.debug_math(2, x)
    assert (dev == oracle.portable_exp(x)).all()
    # ... and the kernels' exp IS the host libm's exp (glibc >= 2.28, FMA variant: the GPU boxes' image) wherever the
    # clamp at -500 is not in the way: the reference's std::exp, bit for bit (round 5), 6e6 arguments
    ref = oracle.libm_exp(x)
    if conftest.libm_is_restated(oracle):
        m = x > -500.0
        assert (dev[m] == ref[m]).all(), "%d arguments differ from the host libm" % int((dev[m] != ref[m]).sum())


# ---------------------------------------------------------------------------
# opt-in fast arithmetic (PMAF_FLAG_FAST_MATH): tolerance parity only
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("cfg,ticks", [("C1", 40), ("C2", 40), ("C3", 3)])
def test_fast_math_within_north_star_tolerance(pmaf, oracle, scenes, cfg, ticks, mode):
    """rcp/rsq + Newton arithmetic (1-2 ulp per op) instead of IEEE div/sqrt:
    not bit-exact by construction. The SELECTED trajectory (the real agent's
    path and the winning agent's predicted path) must stay within 1e-5 m of
    BOTH oracle modes (north star). Individual non-selected agents may diverge
    at C3 (500 steps through 128 obstacles is chaotic: any last-bit
    perturbation, including another libm's exp, is amplified); their share is
    printed. Agent indices are compared through their cost (exactly tied
    agents may swap)."""
    oracle.set_exp_mode(mode)
    sc = scenes.config_scene(cfg)
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"], fast_math=True)
    ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    ora.set_initial_position(sc["start"])
    bh, ph = drive(hip, sc, ticks)
    bo, po = drive(ora, sc, ticks)
    hip.stop()
    assert np.abs(ph - po).max() <= LIBM_TOL
    pth, nh = hip.paths()
    pto, no = ora.paths()
    co = ora.costs()
    assert np.all(np.abs(co[bh[-1]] - co[bo[-1]]) <= 1e-9 * max(1.0, abs(co[bo[-1]])))
    best = int(bo[-1])
    per_agent = np.array([np.abs(pth[a, :min(nh[a], no[a])] - pto[a, :min(nh[a], no[a])]).max() for a in range(len(nh))])
    assert per_agent[best] <= LIBM_TOL
    if cfg != "C3":
        np.testing.assert_array_equal(nh, no)
        assert per_agent.max() <= LIBM_TOL
    print("%s fast-math vs oracle(mode %d): best agent %.3g m, all agents max %.3g m, %d/%d agents within 1e-5 m"
          % (cfg, mode, per_agent[best], per_agent.max(), int((per_agent <= LIBM_TOL).sum()), len(nh)))
    hip.close()


def test_generic_kernel_forced_for_wave_per_agent_mapping(pmaf, oracle, scenes, monkeypatch):
    """PMAF_FORCE_GENERIC=1 keeps k_rollout<64> (the LDS-table kernel) covered
    now that lanes_per_agent = 64 normally dispatches to k_rollout_w64"""
    monkeypatch.setenv("PMAF_FORCE_GENERIC", "1")
    sc = scenes.static1_scene(13, 120)
    hip, _ = run_both(pmaf, oracle, scenes, sc, 6, lanes_per_agent=64)
    hip.close()


# ---------------------------------------------------------------------------
# BASELINE full sizes
# ---------------------------------------------------------------------------
def test_c5_full_size_8x1024_agents_bit_exact(pmaf, oracle, scenes):
    """BASELINE config C5 at full size on one GPU: 8 scenes x 1024 agents x
    200 steps x 32 obstacles in one handle, two ticks, every path point of
    all 8192 agents compared with 8 oracles (bit-exact)."""
    scs = [scenes.config_scene("C5", scene_id=s) for s in range(8)]
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    obs = np.stack([s["obstacles"] for s in scs])
    sc = scs[0]
    for t in range(2):
        bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = [o.tick(obs[i], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for i, o in enumerate(oras)]
        np.testing.assert_array_equal(bh, bo)
    ph, nh = hip.paths()
    for i, o in enumerate(oras):
        po, no = o.paths()
        np.testing.assert_array_equal(nh[i], no)
        np.testing.assert_array_equal(ph[i], po)
        np.testing.assert_array_equal(hip.costs()[i], o.costs())
    # size-independent invariants on the full batch
    vmax, dt = sc["velocity_max"], sc["dt"]
    seg = np.linalg.norm(np.diff(ph, axis=2), axis=3)
    valid = np.arange(1, ph.shape[2])[None, None, :] < nh[:, :, None]
    assert (seg[valid] <= vmax * dt + 0.5 * 13.0 * dt * dt + 1e-12).all()
    mo = hip.min_obs_dist()
    assert (mo >= 1e-5).all() and (mo <= sc["detect_shell_rad"]).all()
    hip.close()


@pytest.mark.parametrize("lpa", [8, 16])
def test_c5_reduced_generic_lane_mappings(pmaf, oracle, scenes, lpa):
    """the throughput mappings (several agents per wave) on a C5-shaped batch"""
    scs = [scenes.synthetic_scene(200, 200, 32, 5, s) for s in range(3)]
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts, lanes_per_agent=lpa)
    hip.set_initial_position(starts)
    obs = np.stack([s["obstacles"] for s in scs])
    sc = scs[0]
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    for t in range(3):
        bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = [o.tick(obs[i], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for i, o in enumerate(oras)]
        np.testing.assert_array_equal(bh, bo)
    ph, nh = hip.paths()
    for i, o in enumerate(oras):
        po, no = o.paths()
        np.testing.assert_array_equal(nh[i], no)
        np.testing.assert_array_equal(ph[i], po)
    hip.close()


def test_winner_records_written_on_device(pmaf, oracle, scenes):
    """pmaf_write_winner_records (send buffer of the all-gather) against the
    host getters; needs torch only to own the device buffer"""
    torch = pytest.importorskip("torch")
    scs = [scenes.synthetic_scene(24, 90, 12, 5, s) for s in range(3)]
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    sc = scs[0]
    obs = np.stack([s["obstacles"] for s in scs])
    for t in range(4):
        hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    best = hip.evaluate(sc["cost_gains"], sc["ws_limits"])
    rec = hip.winner_record_doubles()
    buf = torch.zeros((3, rec), dtype=torch.float64, device="cuda:0")
    hip.write_winner_records(buf.data_ptr(), buf.numel() * 8)
    hip.stop()
    out = pmaf.shard.unpack_winner_records(buf.cpu().numpy(), sc["max_prediction_steps"])
    paths, n = hip.paths()
    costs = hip.costs()
    for p in range(3):
        assert out[p]["index"] == best[p] and out[p]["n_points"] == n[p, best[p]]
        assert out[p]["cost"] == costs[p, best[p]]
        np.testing.assert_array_equal(out[p]["path"], paths[p, best[p], :n[p, best[p]]])
    hip.close()


def test_c4_dual_arm_two_coupled_populations(pmaf, oracle, scenes):
    """BASELINE config 4 on one GPU: 2 x 256 agents, each arm's repulsive
    self-collision sphere tracks the other arm's end effector
    (shard.DualArmCoupling); both arms against two coupled oracles, bit-exact."""
    scs = []
    for arm, (y0, y1) in enumerate(((-0.12, 0.10), (0.12, -0.10))):
        s = scenes.synthetic_scene(256, 150, 24, 4, arm)
        s["start"] = np.array([-0.45, y0, 0.7])
        s["goal"] = np.array([0.45, y1, 0.7])  # the arms' paths cross -> the spheres come into range
        scs.append(s)
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    ch = pmaf.shard.DualArmCoupling(np.stack([s["obstacles"] for s in scs]), 0.1)
    co = pmaf.shard.DualArmCoupling(np.stack([s["obstacles"] for s in scs]), 0.1)
    sc = scs[0]
    pos_h, pos_o = starts.copy(), starts.copy()
    min_gap = 1e9
    for t in range(260):
        oh = ch.coupled_obstacles(pos_h)
        oo = co.coupled_obstacles(pos_o)
        bh = hip.tick(oh, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = [o.tick(oo[i], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for i, o in enumerate(oras)]
        np.testing.assert_array_equal(bh, bo)
        pos_h = hip.real_state()[0].copy()
        pos_o = np.stack([o.real_state()[0] for o in oras])
        np.testing.assert_array_equal(pos_h, pos_o)
        min_gap = min(min_gap, np.linalg.norm(pos_h[0] - pos_h[1]))
    ph, nh = hip.paths()
    for i, o in enumerate(oras):
        po, no = o.paths()
        np.testing.assert_array_equal(nh[i], no)
        np.testing.assert_array_equal(ph[i], po)
    # the coupling was exercised: the two end effectors came within the detection shell
    assert min_gap < sc["detect_shell_rad"] + 0.15
    print("dual arm: closest approach of the two end effectors %.3f m" % min_gap)
    hip.close()


def test_xact_sequences_match_ieee(pmaf):
    """the default arithmetic policy (MATH_XACT, pmaf_device.hpp): hand-expanded
    sqrt / divide sequences must return the correctly rounded IEEE bits for
    operands with exponents within +-250 (the validated input range keeps the
    path's operands far inside) and handle zeros / infinities / NaN like IEEE."""
    rng = np.random.default_rng(7)
    n = 2_000_000
    a = rng.uniform(1e-12, 50.0, n) * rng.choice([-1.0, 1.0], n)
    b = rng.uniform(1e-12, 50.0, n) * rng.choice([-1.0, 1.0], n)
    with np.errstate(all="ignore"):
        assert (pmaf.debug_math(5, np.abs(a)) == np.sqrt(np.abs(a))).all()
        assert (pmaf.debug_math(6, a, b) == a / b).all()
        assert (pmaf.debug_math(7, a, b) == pmaf.debug_math(8, a, b)).all()
        # exponent sweep over the guaranteed range
        wa = np.ldexp(rng.uniform(1.0, 2.0, n), rng.integers(-250, 250, n)) * rng.choice([-1.0, 1.0], n)
        wb = np.ldexp(rng.uniform(1.0, 2.0, n), rng.integers(-250, 250, n)) * rng.choice([-1.0, 1.0], n)
        assert (pmaf.debug_math(5, np.abs(wa)) == np.sqrt(np.abs(wa))).all()
        assert (pmaf.debug_math(6, wa, wb) == wa / wb).all()
        # a / sqrt(b) with the reciprocal of the norm taken from the sqrt iteration (op 9)
        assert (pmaf.debug_math(9, a, np.abs(b)) == a / np.sqrt(np.abs(b))).all()
        assert (pmaf.debug_math(9, wa, np.abs(wb)) == wa / np.sqrt(np.abs(wb))).all()
        v = rng.uniform(-1.5, 1.5, (n, 3))
        z = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
        assert (pmaf.debug_math(9, v[:, 1], z) == v[:, 1] / np.sqrt(z)).all()
        # divisors within a few ulps of 1 (norms of cross products of unit vectors) and numerators on rounding
        # ties: where a reciprocal that is 1 ulp off its correctly rounded value shows (a reciprocal taken from the
        # sqrt iteration failed exactly here: -2^-55 / (1 - 2^-53))
        k = np.arange(-64, 65, dtype=np.float64)
        for base in (1.0, 0.5, 2.0):
            sN = base * (1.0 + k * 2.0 ** -53)
            for num in (2.0 ** -55, 3.0 * 2.0 ** -55, 1.0, 0.3, 1.0 + 2.0 ** -52, -(2.0 ** -55), -0.7):
                aN = np.full_like(sN, num)
                assert (pmaf.debug_math(6, aN, sN) == aN / sN).all()
                assert (pmaf.debug_math(9, aN, sN * sN) == aN / np.sqrt(sN * sN)).all()
        # the classic exception of reciprocal-based division: a divisor whose mantissa is all ones (and its
        # neighbours), against many numerators
        for sN in (1.0 - 2.0 ** -53, 1.0 - 2.0 ** -52, 1.0 + 2.0 ** -52, 2.0 - 2.0 ** -52, 0.5 - 2.0 ** -54):
            aN = np.concatenate([rng.uniform(-3.0, 3.0, 200_000), np.ldexp(rng.integers(1, 1 << 20, 100_000).astype(np.float64), -60),
                                 np.ldexp(1.0, rng.integers(-80, 10, 1000))])
            bN = np.full_like(aN, sN)
            assert (pmaf.debug_math(6, aN, bN) == aN / bN).all()
            assert (pmaf.debug_math(9, aN, bN * bN) == aN / np.sqrt(bN * bN)).all()
        # square roots next to exact squares and to powers of two (results on or beside rounding boundaries)
        kk = np.arange(-200, 201, dtype=np.float64)
        for base in (1.0, 2.0, 0.5, 3.0, 0.1, 7.25):
            r0 = base * (1.0 + kk * 2.0 ** -52)
            for zN in (r0 * r0, np.nextafter(r0 * r0, np.inf), np.nextafter(r0 * r0, 0.0), r0):
                assert (pmaf.debug_math(5, zN) == np.sqrt(zN)).all()
        # the fixup-free variants (ops 11 / 12: divisor a positive normal -- a norm behind a squaredNorm > 0
        # select, a clamped squared distance): same bits as IEEE incl. the sign of a zero numerator
        pb = np.abs(b)
        for (x, y) in ((a, pb), (wa, np.abs(wb)), (np.where(rng.uniform(size=n) < 0.01, -0.0, a), pb)):
            got = pmaf.debug_math(11, x, y)
            assert (got == x / y).all() and (np.signbit(got) == np.signbit(x / y)).all()
            got = pmaf.debug_math(12, x, y)
            assert (got == x / np.sqrt(y)).all() and (np.signbit(got) == np.signbit(x / np.sqrt(y))).all()
        zz = np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -300, -(2.0 ** -300)])
        for y in (1.0, 2.0 ** -200, 3.7, 2.0 ** 200):
            got = pmaf.debug_math(11, zz, np.full_like(zz, y))
            assert (got == zz / y).all() and (np.signbit(got) == np.signbit(zz / y)).all()
        assert (pmaf.debug_math(12, v[:, 1], z) == v[:, 1] / np.sqrt(z)).all()
        for base in (1.0, 0.5, 2.0):
            sN = base * (1.0 + k * 2.0 ** -53)
            for num in (2.0 ** -55, 3.0 * 2.0 ** -55, 1.0, 0.3, 1.0 + 2.0 ** -52, -(2.0 ** -55), -0.7):
                aN = np.full_like(sN, num)
                assert (pmaf.debug_math(11, aN, sN) == aN / sN).all()
                assert (pmaf.debug_math(12, aN, sN * sN) == aN / np.sqrt(sN * sN)).all()
        assert np.isnan(pmaf.debug_math(11, np.array([np.nan]), np.array([2.0]))).all()
        special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.0, 2.0 ** -250, 2.0 ** 250, -7.5])
        A, B = [x.ravel() for x in np.meshgrid(special, special)]
        for op, ref in ((5, np.sqrt(A)), (6, A / B), (9, A / np.sqrt(B))):
            got = pmaf.debug_math(op, A, B)
            same = (got == ref) | (np.isnan(got) & np.isnan(ref))
            assert same.all(), (op, A[~same], B[~same], got[~same], ref[~same])
            assert (np.signbit(got) == np.signbit(ref))[~np.isnan(ref)].all()  # signed zeros / infinities


@pytest.mark.parametrize("cfg,ticks", [("C1", 30), ("C2", 20), ("C3", 2)])
def test_compiler_ieee_sequences_flag_gives_the_same_bits(pmaf, oracle, scenes, cfg, ticks):
    """PMAF_FLAG_IEEE_SEQUENCES (fully general compiler expansions) against the
    oracle, bit-exact like the default policy"""
    sc = scenes.config_scene(cfg)
    hip, _ = run_both(pmaf, oracle, scenes, sc, ticks, ieee_sequences=True)
    hip.close()


def test_inputs_outside_the_supported_numeric_range_are_rejected(pmaf, scenes):
    sc = scenes.config_scene("C1")
    bad = dict(sc)
    bad["obstacles"] = sc["obstacles"].copy()
    bad["obstacles"][2, 1] = 1e-200
    with pytest.raises(pmaf.PmafError) as e:
        pmaf.PmafPlanner(bad, device=0)
    assert e.value.code == -1 and "numeric range" in str(e.value)
    hip = pmaf.PmafPlanner(sc, device=0)
    with pytest.raises(pmaf.PmafError):
        hip.set_initial_position([np.nan, 0.0, 0.0])
    with pytest.raises(pmaf.PmafError):
        hip.set_initial_position([1e40, 0.0, 0.0])
    hip.close()


@pytest.mark.parametrize("one_wave", [False, True])
@pytest.mark.parametrize("m,lpa", [(200, 0), (300, 0), (70, 32), (40, 8)])
def test_large_and_ragged_obstacle_counts(pmaf, oracle, scenes, m, lpa, monkeypatch, one_wave):
    """obstacle counts that exercise 4 slots per lane (M = 200: w64 TILES 4;
    M = 70 @ 32 lanes and M = 40 @ 8 lanes: group kernel TILES 4 / generic),
    the generic fallback (M = 300 > 4 x 64) and ragged last tiles"""
    if one_wave:   # the one-wave kernels (2 / 4 obstacle slots per lane) instead of k_rollout_mw
        monkeypatch.setenv("PMAF_MW", "0")
    sc = scenes.synthetic_scene(20, 120, m, 6, m)
    hip, _ = run_both(pmaf, oracle, scenes, sc, 3, lanes_per_agent=lpa)
    hip.close()


def test_hip_path_reproduces_the_survey_probe_record(pmaf, scenes):
    """the only reference-derived numbers available (SURVEY.md 8c probe record,
    tests/golden/survey_probe.json: the reference's own cf_agent.cpp /
    cf_manager.cpp run by the survey) against the HIP path directly"""
    import json
    import os
    from conftest import std_mt19937_unit_vectors
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_probe.json")))
    N = 10
    rv = np.zeros((N, 10, 3))
    rv[5:] = std_mt19937_unit_vectors(12345, (N - 5) * 10).reshape(N - 5, 10, 3)
    sc = scenes.static1_scene(N, 100, random_vecs=rv)
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    for rec in gold["probe1"]["ticks"]:
        b = hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        assert b == rec["best_index"] and hip.best_type() == rec["best_type"]
        np.testing.assert_allclose(hip.real_state()[0], rec["next"], rtol=0, atol=6e-10)
    assert abs(hip.path_lengths()[0] - gold["probe1"]["agent0_path_length_after_tick2"]) < 6e-10
    hip.close()
    # dyn1 closed loop: goal reached at the probe's tick
    rv = np.zeros((N, 4, 3))
    rv[5:] = std_mt19937_unit_vectors(12345, (N - 5) * 4).reshape(N - 5, 4, 3)
    sc = scenes.dyn1_scene(N, 1500, random_vecs=rv)
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    best, pos = drive(hip, sc, 2000, True, scenes.advance_live_obstacles, until_reached=True)
    assert len(best) - 1 == gold["probe2"]["reached_tick"]
    final = np.asarray(gold["probe2"]["final_real_position"])
    assert np.abs(pos[-1] - final).max() < 1e-3
    hip.close()


def _task_records():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "task_scenes.json")))


@pytest.mark.parametrize("task", sorted(_task_records()))
def test_shipped_task_scenes_closed_loop(pmaf, oracle, scenes, task):
    """every task scene the reference ships (tests/golden/task_scenes.json),
    as shipped: 10 agents, max_prediction_steps 1500 / 1200, obstacle stream
    advancing like dynamic_obstacle_node, planned until getDistFromGoal() < 0.01
    (or 900 ticks); the whole set-point sequence bit-exact against the oracle"""
    sc = scenes.scene_from_record(_task_records()[task], task)
    hip, ora = make_pair(pmaf, oracle, sc)
    bh, ph = drive(hip, sc, 900, True, scenes.advance_live_obstacles, until_reached=True)
    bo, po = drive(ora, sc, 900, True, scenes.advance_live_obstacles, until_reached=True)
    hip.stop()
    assert len(bh) == len(bo)
    np.testing.assert_array_equal(bh, bo)
    np.testing.assert_array_equal(ph, po)
    assert_state_equal(hip, ora)
    print("%s: %d ticks, final goal distance %.4f, best-agent switches %d" %
          (task, len(bh), hip.dist_from_goal(), int((np.diff(bh) != 0).sum())))
    hip.close()


@pytest.mark.parametrize("task", sorted(_task_records()))
def test_shipped_task_scenes_against_libm_exp_oracle(pmaf, oracle, scenes, task, request):
    """The north star's 1e-5 m at the reference's OWN operating point: every
    shipped task scene as shipped (H = 1500 / 1200, moving obstacles, closed loop
    until reached / 900 ticks), HIP path (portable exp) against the oracle in
    its reference-faithful mode (mode 0: the platform libm's exp, as
    cf_agent.cpp:220 calls it). Per scene: the set-point sequence, the first
    tick at which the best index differs (none), and the deviation of the
    SELECTED trajectory (the best agent's predicted path that was scored)."""
    conftest.expect_chaotic(request, "task_libm:" + task)
    oracle.set_exp_mode(0)
    LIBM_TOL = conftest.libm_tol(oracle, 1e-5)
    sc = scenes.scene_from_record(_task_records()[task], task)
    hip, ora = make_pair(pmaf, oracle, sc)
    obs = sc["obstacles"].copy()
    max_set, max_sel, first_flip, n_ticks = 0.0, 0.0, None, 0
    for t in range(900):
        hip.stop()
        ph, nh = hip.paths()
        po, no = ora.paths()
        bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        obs = scenes.advance_live_obstacles(obs)
        n_ticks += 1
        if bh != bo and first_flip is None:
            first_flip = t
        if first_flip is None:
            assert nh[bh] == no[bo]
            max_sel = max(max_sel, float(np.abs(ph[bh, :nh[bh]] - po[bo, :no[bo]]).max()))
        max_set = max(max_set, float(np.abs(hip.real_state()[0] - ora.real_state()[0]).max()))
        if hip.dist_from_goal() < 0.01 and ora.dist_from_goal() < 0.01:
            break
    print("%-24s %4d ticks | set-point dev %.3g m | selected-trajectory dev %.3g m | first best-index difference: %s"
          % (task, n_ticks, max_set, max_sel, first_flip))
    assert first_flip is None
    assert max_set <= LIBM_TOL and max_sel <= LIBM_TOL
    hip.close()


def test_checkpoint_resume_is_bit_identical(pmaf, oracle, scenes):
    """pmaf_save_state / pmaf_load_state: a planner restored from a blob (into a
    fresh handle) continues exactly like the original and like the oracle"""
    sc = scenes.config_scene("C2", scene_id=2, dynamic=True)
    hip, ora = make_pair(pmaf, oracle, sc)
    obs = sc["obstacles"].copy()
    for t in range(15):
        assert hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]) == ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        obs = scenes.advance_live_obstacles(obs)
    blob = hip.save_state()
    obs_at_save = obs.copy()
    ref = []
    for t in range(12):
        b = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        assert b == ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        ref.append((b, hip.real_state()[0].copy()))
        obs = scenes.advance_live_obstacles(obs)
    hip.stop()
    paths_ref, n_ref = hip.paths()
    traj_ref = hip.real_path()
    # restore into a brand-new handle (different initial contents) and replay
    other = dict(sc)
    other["random_vecs"] = sc["random_vecs"][::-1].copy()
    other["goal"] = sc["goal"] + 0.1          # everything comes from the blob, incl. goal and scalars
    other["velocity_max"] = 0.3
    hip2 = pmaf.PmafPlanner(other, device=0)
    hip2.load_state(blob)
    obs = obs_at_save
    for t in range(12):
        b = hip2.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        assert b == ref[t][0]
        np.testing.assert_array_equal(hip2.real_state()[0], ref[t][1])
        obs = scenes.advance_live_obstacles(obs)
    hip2.stop()
    p2, n2 = hip2.paths()
    np.testing.assert_array_equal(n2, n_ref)
    np.testing.assert_array_equal(p2, paths_ref)
    np.testing.assert_array_equal(hip2.real_path(), traj_ref)
    assert hip2.dist_from_goal() == ora.dist_from_goal()
    assert_state_equal(hip2, ora)
    # a blob from a differently sized handle is refused
    small = pmaf.PmafPlanner(scenes.config_scene("C1"), device=0)
    with pytest.raises(pmaf.PmafError):
        small.load_state(blob)
    small.close(); hip.close(); hip2.close()


@pytest.mark.parametrize("cuts", [[0, 8, 24], [0, 5, 11, 17, 24]])
def test_agent_range_sharding_equals_single_population(pmaf, oracle, scenes, cuts):
    """one population split over 2 / 4 shards (SURVEY 8e fallback; here all
    shards live in one process, the collective is the identity): global
    selection through merge_agent_ranges + a replicated real agent must give the
    unsharded planner's (= the oracle's) set-points and paths bit for bit. The
    shipped static1 scene with 24 agents switches its best agent twice in the
    first 40 ticks (Had -> Goal -> a Random agent of a later shard)."""
    rec = dict(_task_records()["dual_arms_static1"])
    rec["n_agents"] = 24
    sc = scenes.scene_from_record(rec, "static1_24", horizon=400)
    shards = [pmaf.shard.AgentRangeShard(pmaf.PmafPlanner, sc, a, b, device=0, mgr_init_pos=sc["start"])
              for a, b in zip(cuts[:-1], cuts[1:])]
    for s in shards:
        s.planner.set_initial_position(sc["start"])
    ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
    ora.set_initial_position(sc["start"])
    prev = None
    seen = set()
    for t in range(40):
        best, pos = pmaf.shard.sharded_tick(shards, prev, sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        assert best == bo
        np.testing.assert_array_equal(pos, ora.real_state()[0])
        seen.add(best)
        prev = best
    assert len(seen) >= 3  # hysteresis switches across shard boundaries were exercised
    po, no = ora.paths()
    for s in shards:
        s.planner.stop()
        ph, nh = s.planner.paths()
        np.testing.assert_array_equal(nh, no[s.a0:s.a1])
        np.testing.assert_array_equal(ph, po[s.a0:s.a1])
        s.planner.close()


@pytest.mark.parametrize("one_wave", [False, True])
@pytest.mark.parametrize("lpa,force_generic", [(0, False), (16, False), (8, False), (0, True)])
def test_closest_other_ties_and_far_obstacles(pmaf, oracle, scenes, monkeypatch, lpa, force_generic, one_wave):
    """the Obstacle / GoalObstacle heuristics latch against the nearest OTHER
    obstacle (cf_agent.cpp:434-446, :480-492: ascending scan, strict `>`, 100 m
    initial minimum). The tuned kernels do that scan cooperatively over the
    lanes, so pin its corner cases against the oracle's sequential scan:
    exact distance ties (lowest index wins, across lanes and across slots of one
    lane), a neighbour exactly 100 m away (not accepted) and obstacles with no
    neighbour inside 100 m (index 0, which may be the obstacle itself)."""
    if one_wave:   # the one-wave kernels (2 / 4 obstacle slots per lane) instead of k_rollout_mw
        monkeypatch.setenv("PMAF_MW", "0")
    if force_generic:
        monkeypatch.setenv("PMAF_FORCE_GENERIC", "1")
    types = np.array([2, 3, 2, 3, 5, 6, 1, 4], dtype=np.int32)
    # field obstacles on the agents' way, symmetric about the centre one: |o1-o0| == |o1-o2| etc.
    near = [[0.00, 0.02, 0.70], [0.10, 0.02, 0.70], [-0.10, 0.02, 0.70], [0.00, 0.12, 0.70], [0.00, -0.08, 0.70],
            [0.00, 0.02, 0.80], [0.00, 0.02, 0.60]]
    m = 70  # two slots per lane in the w64 mapping, five in the 16-lane one
    sc = scenes.synthetic_scene(8, 150, m, 9, 5, agent_types=types)
    obs = sc["obstacles"]
    for k in range(m):
        obs[k, :3] = near[k] if k < len(near) else [150.0 + 7.0 * k, 40.0, 0.7]  # off the path
        obs[k, 3:6] = 0.0
        obs[k, 6] = 0.03
    # a pair exactly 100 m apart and one obstacle alone in its neighbourhood (> 100 m from everything)
    obs[m - 3, :3] = [0.3, 0.0, 0.7]
    obs[m - 2, :3] = [0.3, 100.0, 0.7]
    obs[m - 1, :3] = [-0.3, -0.05, 0.72]
    hip, _ = run_both(pmaf, oracle, scenes, sc, 4, lanes_per_agent=lpa)
    assert hip.known().any()  # latches happened
    hip.close()
    # a sparse scene: every field obstacle is > 100 m from every other one except those on the path
    sc = scenes.synthetic_scene(4, 150, 3, 9, 6, agent_types=types[:4])
    sc["obstacles"][0] = [0.0, 250.0, 0.7, 0, 0, 0, 0.03]
    sc["obstacles"][1] = [0.05, 0.03125, 0.7, 0, 0, 0, 0.03]
    sc["obstacles"][2] = [0.05, -99.96875, 0.7, 0, 0, 0, 0.03]   # exactly 100 m from obstacle 1: `100 > d` is false
    hip, _ = run_both(pmaf, oracle, scenes, sc, 3, lanes_per_agent=lpa)
    hip.close()


def test_prediction_times_are_per_agent(pmaf, scenes):
    """getPredictionTimes (cf_manager.cpp:200-206): one duration per agent from
    the device clock; agents that stop at the goal after a few steps report
    shorter rollouts than agents that run the whole horizon"""
    sc = scenes.config_scene("C2")
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    hip.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    t = np.asarray(hip.prediction_times_ns()).reshape(-1)
    assert t.shape == (sc["n_agents"],)
    assert (t > 2e4).all() and (t < 2e7).all()       # 200 steps: tens of us .. well under 20 ms
    assert np.unique(t).size > 8                      # per agent, not one launch time
    hip.close()
    near = dict(sc)
    near["start"] = sc["goal"] + np.array([-0.12, 0.0, 0.0])   # guard ends after a few steps
    hip = pmaf.PmafPlanner(near, device=0, mgr_init_pos=near["start"])
    hip.set_initial_position(near["start"])
    hip.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    assert (np.asarray(hip.n_points()) < 60).all()
    assert np.asarray(hip.prediction_times_ns()).max() < 0.5 * t.min()
    hip.close()


@pytest.mark.parametrize("lpa", [0, 16])
def test_signed_zero_coordinates_of_obstacles_at_rest(pmaf, oracle, scenes, lpa):
    """obstacles at rest are advanced once per rollout by the tuned kernels
    (p + (+-0) dt is idempotent after the first application): -0.0 coordinates
    and -0.0 velocity components must still behave like the reference's
    `pos += vel * dt` in every step (a -0.0 coordinate turns into +0.0 after the
    first step unless its velocity component is -0.0 too)"""
    sc = scenes.synthetic_scene(12, 150, 6, 9, 7)
    obs = sc["obstacles"]
    obs[0, :3] = [0.0, -0.0, 0.7]      # on the start-goal line (y = z-offset 0): zeros everywhere
    obs[1, :3] = [-0.0, 0.05, 0.7]
    obs[1, 3:6] = [-0.0, 0.0, -0.0]
    obs[2, :3] = [0.2, -0.0, 0.7]
    obs[2, 3:6] = [0.0, -0.0, 0.0]
    hip, _ = run_both(pmaf, oracle, scenes, sc, 4, lanes_per_agent=lpa)
    hip.close()


@pytest.mark.parametrize("one_wave", [False, True])
@pytest.mark.parametrize("m", [59, 60, 61, 62, 63, 64, 65])
@pytest.mark.parametrize("dynamic", [False, True])
def test_idle_lane_riders_at_the_obstacle_count_boundary(pmaf, oracle, scenes, m, dynamic, monkeypatch, one_wave):
    """the one-slot wave-per-agent kernel runs the goal distance / direction, the
    speed clamp and the attractor speed limit in lanes 63 / 62 / 61 of the sweep's
    norm sequence (and a reachable repulsive obstacle in lane 60), so it takes at
    most 60 field obstacles; 61...64 go to the split / two-slot kernels (which pack
    the three riders into a sequence of their own)"""
    if one_wave:   # the one-wave kernels (2 / 4 obstacle slots per lane) instead of k_rollout_mw
        monkeypatch.setenv("PMAF_MW", "0")
    sc = scenes.synthetic_scene(12, 90, m, 6, 100 + m, dynamic=dynamic)
    hip, _ = run_both(pmaf, oracle, scenes, sc, 3, dynamic=dynamic, lanes_per_agent=64)
    assert hip.launch_config()["lanes_per_agent"] == 64
    hip.close()


def test_idle_lane_goal_with_signed_zero_coordinates(pmaf, oracle, scenes):
    """lane 63 of the one-slot kernel carries the goal like an obstacle at rest,
    with velocity -0.0 so that the update `p + v dt` leaves a -0.0 coordinate as
    it is (with +0.0 it would turn into +0.0 after the first step). The goal
    direction read back from that lane also feeds the first-contact latch
    (calc_rot_vec_pre: the Random heuristic's rotation vector is
    cross(goal direction, random vector)), so the rotation vectors are compared
    BIT for bit here, signs of zeros included. Start and goal on the x axis
    (y = -0.0 / +0.0, z equal): zeros in g.y all the way."""
    for goal_y, start_y in ((-0.0, 0.0), (0.0, -0.0), (-0.0, -0.0)):
        sc = scenes.synthetic_scene(12, 120, 9, 6, 31)
        sc["goal"] = np.array([0.6, goal_y, 0.7])
        sc["start"] = np.array([-0.6, start_y, 0.7])
        sc["obstacles"][:3, 1] = [0.0, -0.0, 0.0]      # field obstacles on the axis too: y stays an exact zero
        sc["obstacles"][:3, 2] = 0.7
        sc["obstacles"][3:9, 0] += 10.0                # the others far away
        for dyn in (False, True):
            hip, ora = run_both(pmaf, oracle, scenes, sc, 3, dynamic=dyn, lanes_per_agent=64)
            rh, ro = np.ascontiguousarray(hip.rot_vecs()), np.ascontiguousarray(ora.rot_vecs())
            assert hip.known().sum() > 0          # some obstacle was latched
            np.testing.assert_array_equal(rh.view(np.uint64), ro.view(np.uint64))
            hip.close()


def test_repulsive_obstacle_moving_into_range(pmaf, oracle, scenes):
    """the repulsive (last) obstacle starts out of range and flies towards the
    agents: the once-per-rollout reachability bound must keep the per-step range
    test alive (repelForce, cf_agent.cpp:159-181), on every kernel mapping"""
    for lpa in (0, 16):
        sc = scenes.synthetic_scene(8, 200, 5, 9, 8, dynamic=True)
        # 2 m away, 1.2 m/s towards the path: its surface comes within 0.27 m of the predicted paths (shell 0.35 m)
        sc["obstacles"][-1] = [0.0, 2.0, 0.7, 0.0, -1.2, 0.0, 0.1]
        hip, ora = run_both(pmaf, oracle, scenes, sc, 3, dynamic=True, lanes_per_agent=lpa)
        hip.close()


@pytest.mark.parametrize("m", [32, 59, 60, 61, 62])
@pytest.mark.parametrize("case", ["flying_in", "resting_on_the_path", "on_an_axis", "fast", "ieee"])
def test_repulsive_obstacle_rides_in_lane_60(pmaf, oracle, scenes, m, case):
    """the one-slot kernel's loop for a reachable repulsive obstacle keeps it in lane 60 like a field obstacle (advanced by
    the same p + v dt) and takes |p - sent_pos| and the direction from the tail's norm sequence, so it holds at most 60
    field obstacles -- 61 and more go to the split / two-slot kernels, which evaluate repelForce on their own. Around
    that boundary: the obstacle flying in (out of range, in range, out again), resting right next to the path (in range
    from the first step), centred on a coordinate axis of the start (zero components of the direction: their sign
    differs from the reference's normalized(p - sent_pos) and F + (0 + repel) must not see it), and the other
    arithmetic policies' kernels"""
    sc = scenes.synthetic_scene(9, 160, m, 9, 300 + m, dynamic=(case == "flying_in"))
    if case == "flying_in":
        sc["obstacles"][-1] = [0.0, 2.0, 0.7, 0.0, -1.2, 0.0, 0.1]
    elif case == "on_an_axis":
        sc["obstacles"][-1] = [sc["start"][0], sc["start"][1] + 0.3, sc["start"][2], 0.0, 0.0, 0.0, 0.08]
    else:
        sc["obstacles"][-1] = [-0.2, 0.12, 0.72, 0.0, 0.0, 0.0, 0.1]
    kw = {"fast_math": True} if case == "fast" else {"ieee_sequences": True} if case == "ieee" else {}
    if case == "fast":
        # (1-2 ulp arithmetic: tolerance parity as in test_fast_math_within_north_star_tolerance, selected trajectory)
        hip, ora = make_pair(pmaf, oracle, sc, **kw)
        for _ in range(3):
            b = ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
            assert hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"]) == b
        hip.stop()
        (ph, nh), (po, no) = hip.paths(), ora.paths()
        assert nh[b] == no[b] and np.abs(ph[b, :no[b]] - po[b, :no[b]]).max() <= LIBM_TOL
        hip.close()
        return
    hip, _ = run_both(pmaf, oracle, scenes, sc, 3, dynamic=(case == "flying_in"), **kw)
    cfg = hip.launch_config()
    assert (cfg["waves_per_agent"] == 1) == (m <= 60 or case == "ieee")
    hip.close()


@pytest.mark.parametrize("lpa,force_generic,ieee", [(0, False, False), (0, True, False), (0, False, True)])
def test_live_radius_of_the_repulsive_obstacle_differs_from_init(pmaf, oracle, scenes, monkeypatch, lpa, force_generic, ieee):
    """the caller's live obstacle list may carry another radius for the last
    (repulsive) obstacle than the list given to init -- shard.DualArmCoupling
    rewrites it every tick. The real agent's step (RealCfAgent::cfPlanner ->
    repelForce, cf_agent.cpp:159-181) uses the LIVE radius for both the range
    test and the distance; the agents' private copies keep the init radius
    (setObstacles does not copy it, cf_agent.cpp:63-70). Radii chosen so that the
    live sphere is in range when the init sphere would not be, and vice versa."""
    if force_generic:
        monkeypatch.setenv("PMAF_FORCE_GENERIC", "1")
    sc = scenes.synthetic_scene(12, 120, 8, 9, 3)
    sc["obstacles"][-1] = [-0.6, 0.55, 0.7, 0.0, 0.0, 0.0, 0.1]     # 0.55 m beside the start: out of range at r = 0.1
    hip, ora = make_pair(pmaf, oracle, sc, lanes_per_agent=lpa, ieee_sequences=ieee)
    obs = sc["obstacles"].copy()
    forces = []
    for t in range(60):
        live = obs.copy()
        live[-1, 6] = (0.30, 0.02, 0.18)[t % 3] if t >= 5 else 0.1   # r = 0.30 / 0.18: in range; 0.02: out
        bh = hip.tick(live, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = ora.tick(live, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        assert bh == bo
        for a, b in zip(hip.real_state(), ora.real_state()):
            np.testing.assert_array_equal(a, b)
        forces.append(hip.real_state()[2].copy())
    hip.stop()
    assert_state_equal(hip, ora)
    # the repulsion was switched on and off by the live radius alone
    fy = np.asarray(forces)[:, 1]
    t = np.arange(fy.size)
    assert np.all(fy[(t >= 5) & (t % 3 == 0)] < -0.4)            # r = 0.30: pushed away from the sphere
    assert np.all(np.abs(fy[(t >= 5) & (t % 3 == 1)]) < 0.3)     # r = 0.02: out of range, attractor terms only
    hip.close()


def test_handle_lifecycle_does_not_leak_device_memory(pmaf, scenes):
    """create / tick / destroy in a loop: device memory returns to where it was
    (CfManager's destructor joins its threads and frees everything,
    cf_manager.h:51), and two live handles do not disturb each other"""
    import torch
    sc = scenes.config_scene("C1")

    def one():
        h = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
        h.set_initial_position(sc["start"])
        for _ in range(3):
            h.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop()
        out = np.asarray(h.real_state()[0]).copy()
        h.close()
        return out

    ref = one()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(25):
        np.testing.assert_array_equal(one(), ref)
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < 8 << 20, (free0, free1)   # allocator slack only
    a = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    b = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    a.set_initial_position(sc["start"]); b.set_initial_position(sc["start"])
    for _ in range(3):
        a.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        b.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    a.stop(); b.stop()
    np.testing.assert_array_equal(a.real_state()[0], ref)
    np.testing.assert_array_equal(b.real_state()[0], ref)
    a.close(); b.close()


def test_two_handles_driven_from_two_threads(pmaf, scenes):
    """a handle is not thread-safe (like CfManager), but different handles are
    independent: two host threads tick their own planners concurrently (ctypes
    releases the GIL, each handle has its own stream and mailbox) and get the
    results of a sequential run"""
    import threading
    sca, scb = scenes.config_scene("C2"), scenes.config_scene("C2", scene_id=3)

    def run(sc, out, n=40):
        h = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
        h.set_initial_position(sc["start"])
        best = [np.asarray(h.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])).copy() for _ in range(n)]
        h.stop()
        out.append((np.asarray(best), np.asarray(h.real_state()[0]).copy(), np.asarray(h.costs()).copy()))
        h.close()

    ref_a, ref_b = [], []
    run(sca, ref_a); run(scb, ref_b)
    got_a, got_b = [], []
    ta = threading.Thread(target=run, args=(sca, got_a)); tb = threading.Thread(target=run, args=(scb, got_b))
    ta.start(); tb.start(); ta.join(); tb.join()
    for ref, got in ((ref_a, got_a), (ref_b, got_b)):
        assert len(got) == 1
        for x, y in zip(ref[0], got[0]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("lpa,force_generic", [(0, False), (16, False), (0, True)])
def test_workspace_penalties_of_many_path_points(pmaf, oracle, scenes, monkeypatch, lpa, force_generic):
    """evaluateAgents' workspace-box penalties (cf_manager.cpp:302-324) are
    summed over the path points in order, x / y / z terms within a point; the
    tuned kernels evaluate them after the rollout, 64 (or lanes-per-agent)
    points at a time. A box the paths leave on several sides and in every chunk
    must give the oracle's costs bit for bit (and a different winner than the
    open box)"""
    if force_generic:
        monkeypatch.setenv("PMAF_FORCE_GENERIC", "1")
    sc = scenes.synthetic_scene(24, 220, 12, 9, 9)
    sc["ws_limits"] = np.array([0.25, -0.45, 0.04, -0.02, 0.74, 0.69])   # xmax xmin ymax ymin zmax zmin
    hip, ora = run_both(pmaf, oracle, scenes, sc, 4, lanes_per_agent=lpa)
    assert (np.asarray(ora.costs()) > 10.0).all()     # every agent pays workspace penalties
    hip.close()


def test_randomised_scenes_bit_exact(pmaf):
    """tools/fuzz_parity.py: random agent / obstacle counts (0 ... 150), heuristic mixes, moving obstacles, gains,
    horizons and lanes-per-agent mappings, every result compared bit for bit with the oracle. (20 000 trials of it
    found the one arithmetic shortcut that was not exact: a reciprocal shared with the sqrt iteration.)"""
    import subprocess
    import sys as _sys
    import conftest
    r = subprocess.run([_sys.executable, os.path.join(conftest.ROOT, "tools", "fuzz_parity.py"), "1500", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout


def test_randomised_api_sequences_bit_exact(pmaf):
    """tools/fuzz_api.py: 1..5 populations per handle, ticks issued as pmaf_tick or as the reference's five-call
    sequence, save_state / load_state hand-overs to fresh handles in between; every population against its own
    oracle, bit for bit"""
    import subprocess
    import sys as _sys
    import conftest
    r = subprocess.run([_sys.executable, os.path.join(conftest.ROOT, "tools", "fuzz_api.py"), "300", "9"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout


@pytest.mark.parametrize("lpa", [0, 16, 1])
def test_synchronous_stepping_api_matches_oracle(pmaf, oracle, scenes, lpa):
    """CfManager::moveAgents / moveAgent / setEEAgentPositions /
    setEEAgentPosAndVels and CfAgent::evalObstacleDistance (cf_manager.cpp:220-291,
    cf_agent.cpp:146-157, 278-300) on the device against the oracle's restatement:
    caller-supplied obstacles incl. their radii, no obstacle advance, the call's
    own delta_t, agents continue from their own state; then back to the tick"""
    sc = scenes.synthetic_scene(20, 400, 14, 9, 5, dynamic=True)
    hip, ora = make_pair(pmaf, oracle, sc, lanes_per_agent=lpa)
    obs = sc["obstacles"].copy()
    for t in range(3):                      # some latched rotation vectors / known flags first
        assert hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]) == ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    start = hip.real_state()[0]
    hip.set_agent_pos_and_vels(start, [0.9, 0.1, 0.0])   # clamped to vel_max
    ora.set_agent_pos_and_vels(start, [0.9, 0.1, 0.0])
    assert_state_equal(hip, ora)
    with pytest.raises(pmaf.PmafError) as e:
        hip.move_agents(obs, 0.01, 1000)                 # would outgrow the path buffers
    assert e.value.code == -3
    for k, (dt, steps) in enumerate(((0.01, 40), (0.02, 25), (0.005, 60))):
        live = scenes.advance_live_obstacles(obs, 100.0 / (k + 1))
        live[:-1, 6] *= 1.0 + 0.1 * k                    # the caller's radii count (cfPlanner uses the list as given)
        hip.move_agents(live, dt, steps)
        ora.move_agents(live, dt, steps)
        assert_state_equal(hip, ora)
        np.testing.assert_array_equal(hip.eval_obstacle_distance(live), ora.eval_obstacle_distance(live))
    with pytest.raises(pmaf.PmafError) as e:
        hip.start()                                       # rollouts need a reset after stepping
    assert e.value.code == -3
    # scoring works on the stepped paths (k_score) -- evaluate changes best_agent_ on both sides alike
    assert hip.evaluate(sc["cost_gains"], sc["ws_limits"]) == ora.evaluate(sc["cost_gains"], sc["ws_limits"])
    np.testing.assert_array_equal(hip.costs(), ora.costs())
    # moveAgent: one agent until it is within 0.05 of the goal (or the buffer is full)
    hip.set_agent_positions(sc["goal"] - np.array([0.3, 0.02, 0.0]))
    ora.set_agent_positions(sc["goal"] - np.array([0.3, 0.02, 0.0]))
    room = (sc["max_prediction_steps"] - 1) // 10
    for agent in (7, 1):
        ch = hip.move_agent(obs, 0.01, 10, agent, max_calls=room)
        co = ora.move_agent(obs, 0.01, 10, agent, max_calls=room)
        assert ch == co and 0 < ch
    assert_state_equal(hip, ora)
    assert hip.n_points()[7] > 11 and hip.n_points()[0] == 1
    # and the ordinary tick continues from there, bit for bit
    pos, vel, _ = hip.real_state()
    hip.reset_agents(pos, vel, obs); ora.reset_agents(pos, vel, obs)
    hip.start(); ora.start()
    for t in range(4):
        assert hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]) == ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        obs = scenes.advance_live_obstacles(obs)
    hip.stop()
    assert_state_equal(hip, ora)
    hip.close()


@pytest.mark.parametrize("n,p,m,want", [
    # round 5's rows: never more than two obstacle slots per lane in a narrower mapping
    (2304, 1, 128, 64), (2304, 1, 70, 64), (4096, 1, 64, 32), (4096, 1, 33, 32), (2304, 1, 64, 32), (8192, 1, 32, 16),
    (8400, 1, 20, 16), (12288, 1, 16, 8),
    # round 6 (pmaf_lpa_model.hpp, profiles/r6_lpa_grid.txt): the three regions where the wave-count rule was 20-31 % off ...
    (2304, 1, 9, 16), (3072, 1, 16, 16), (4096, 1, 12, 16),                # few obstacles: 16 lanes, not 32
    (2304, 1, 48, 64), (3072, 1, 60, 64), (6144, 1, 60, 64),               # one-slot wave per agent through a third round
    (1536, 1, 64, 32), (2048, 1, 62, 32),                                  # 61-64 obstacles: 32 lanes x 2 slots at one wave per SIMD
    # ... the same decisions for several populations in one handle (waves = P x ceil(N L / 64)) ...
    (1024, 3, 9, 16), (256, 8, 62, 32), (300, 9, 50, 64),
    # ... and the band between one and two waves per SIMD of the wave per agent, where nothing changes (profiles/r6_lpa_band.txt)
    (1280, 1, 32, 64), (2048, 1, 32, 64), (1024, 2, 32, 64), (1792, 1, 9, 64), (2048, 1, 60, 64)])
def test_narrower_mappings_are_chosen_by_the_measured_table(pmaf, oracle, scenes, n, p, m, want):
    """pick_lpa: the mapping with the smallest estimated launch time (csrc/pmaf_lpa_model.hpp, a table of measured times;
    round 5's structural rule -- at most two obstacle slots per lane in a narrower mapping -- is what it offers from).
    Parity of the populations whose mapping changed in round 6, of round 5's, and of neighbours whose mapping stayed:
    every path point, cost and index against the oracle, tolerance 0, P populations against P oracles."""
    assert pmaf.load_library().pmaf_pick_lanes_per_agent(n, p, m, 0) == want
    scs = [scenes.synthetic_scene(n, 12, m, 6, 9 + i) for i in range(p)]
    sc = scs[0]
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs if p > 1 else sc, device=0, mgr_init_pos=starts if p > 1 else sc["start"])
    hip.set_initial_position(starts if p > 1 else sc["start"])
    assert hip.launch_config()["lanes_per_agent"] == want
    # two one-slot wave-per-agent rollouts per SIMD run k_rollout_w64_sliced (priority slices: scheduling only, same bits)
    assert hip.launch_config()["priority_slices"] == (want == 64 and m <= 60 and 1024 < n * p <= 2048)
    oras = []
    for q in scs:
        o = oracle.OraclePlanner(q, mgr_init_pos=q["start"])
        o.set_initial_position(q["start"])
        oras.append(o)
    obs = np.stack([q["obstacles"] for q in scs])
    for t in range(2):
        bh = np.atleast_1d(hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]))
        bo = [o.tick(obs[i], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for i, o in enumerate(oras)]
        np.testing.assert_array_equal(bh, bo)
    hip.stop()
    if p == 1:
        assert_state_equal(hip, oras[0])
    else:
        ph, nh = hip.paths()
        for i, o in enumerate(oras):
            po, no = o.paths()
            np.testing.assert_array_equal(nh[i], no)
            np.testing.assert_array_equal(ph[i], po)
            np.testing.assert_array_equal(hip.costs()[i], o.costs())
            np.testing.assert_array_equal(hip.rot_vecs()[i], o.rot_vecs())
            np.testing.assert_array_equal(np.asarray(hip.real_state()[0]).reshape(p, 3)[i], o.real_state()[0])
    hip.close()


def _static_lds_of(pmaf, pattern):
    """largest `LDS Size [bytes/block]` the compiler reported for the kernels whose mangled name contains `pattern`
    (csrc/build.sh keeps -Rpass-analysis=kernel-resource-usage in lib/resource_usage.txt)"""
    import re
    f = os.path.join(os.path.dirname(pmaf.LIB_PATH), "resource_usage.txt")
    if not os.path.exists(f):
        pytest.skip("no resource_usage.txt next to the library (built by another recipe)")
    sizes = [int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
             for b in open(f).read().split("Function Name: ")[1:] if pattern in b.split("\n")[0]]
    assert sizes, pattern
    return max(sizes)


@pytest.mark.parametrize("m", [65, 100, 128])
def test_eight_one_wave_blocks_of_the_two_slot_kernel_fit_a_cu(pmaf, scenes, m):
    """2048+ agents x 65..128 obstacles run on k_rollout_w64<2, ...>: one-wave blocks, two waves per SIMD = eight blocks
    per CU -- if static + dynamic LDS of a block stay within 160 KB / 8. Round 5 broke exactly that once (a __shared__
    table per instantiation of the step body + the list area sized for four slots: 21.1 KB, seven blocks per CU, a second
    round of blocks, +45 % per launch; found by tools/regime.py's sweep, not by a test). Sizes as the build reports them
    and as pmaf_create requests them."""
    sc = scenes.synthetic_scene(2304, 8, m, 6, 3)
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    cfg = hip.launch_config()
    hip.close()
    assert cfg["lanes_per_agent"] == 64 and cfg["waves_per_agent"] <= 1
    static = _static_lds_of(pmaf, "k_rollout_w64ILi2E")
    granule = 1280       # allocation granule assumed no coarser than 160 KB / 128
    per_block = -(-(static + cfg["lds_bytes"]) // granule) * granule
    assert 8 * per_block <= 160 * 1024, (static, cfg["lds_bytes"], per_block)


def test_two_handles_with_obstacle_tables_beyond_64_kb(pmaf, oracle, scenes):
    """more than 64 KB of dynamic LDS needs a per-KERNEL opt-in (hipFuncAttributeMaxDynamicSharedMemorySize): a second
    handle with a smaller table (still > 64 KB) must not lower the limit under the first one's launches -- the largest
    request of the process stands (csrc/pmaf_k_misc.hip: pmaf_k_set_lds_limits). 1500 and 1000 obstacles, the larger
    handle created first, ticked alternately. (ROCm 7.2 does not enforce the attribute -- the per-handle setter passed
    this test as well; the rule is kept for runtimes that do.)"""
    scs = [scenes.synthetic_scene(5, 10, 1500, 6, 11), scenes.synthetic_scene(5, 10, 1000, 6, 12)]
    pairs = [make_pair(pmaf, oracle, sc) for sc in scs]
    assert [p[0].launch_config()["lds_bytes"] > 64 * 1024 for p in pairs] == [True, True]
    assert pairs[0][0].launch_config()["lds_bytes"] > pairs[1][0].launch_config()["lds_bytes"]
    for t in range(2):
        for (hip, ora), sc in zip(pairs, scs):
            assert hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"]) == \
                ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    for hip, ora in pairs:
        hip.stop()
        assert_state_equal(hip, ora)
        hip.close()


def test_many_agents_with_200_obstacles_stay_on_the_four_slot_kernel(pmaf, oracle, scenes):
    """N x P > 1024 waves with 129..256 obstacles: the launch keeps the
    wave-per-agent mapping (k_rollout_w64<4>, several rounds of waves) instead of
    falling to the generic kernel (2.7-3.7x slower, tools/m200time.py); parity of
    a 1200-agent population through 200 obstacles"""
    sc = scenes.synthetic_scene(1200, 60, 200, 6, 2)
    hip, ora = make_pair(pmaf, oracle, sc)
    assert hip.launch_config()["lanes_per_agent"] == 64
    for t in range(2):
        assert hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"]) == \
            ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    assert_state_equal(hip, ora)
    hip.close()


@pytest.mark.parametrize("mode", ["lds", "dpp"])
@pytest.mark.parametrize("cfg,ticks,dynamic", [("C1", 12, False), ("C2", 10, True)])
def test_both_ordered_sum_variants_of_the_one_slot_kernel(pmaf, oracle, scenes, monkeypatch, mode, cfg, ticks, dynamic):
    """the ordered force sum of the wave-per-agent kernel exists in two variants
    (LDS batches, the rounds-1/2 choice for short lists, and the DPP row_newbcast chain the
    host picks for every obstacle count since round 3): both on both configs, bit-exact, plus a dense scene
    with up to 61 in-shell terms (4 chunks of 16, chunk-boundary padding)"""
    monkeypatch.setenv("PMAF_SUM", mode)
    sc = scenes.config_scene(cfg, dynamic=dynamic) if cfg != "C1" else scenes.config_scene(cfg)
    hip, _ = run_both(pmaf, oracle, scenes, sc, ticks, dynamic=dynamic)
    hip.close()
    # every obstacle within the shell of the start: lists of 16, 32, 48 and 61 entries
    for m in (16, 32, 48, 61):
        sc = scenes.synthetic_scene(8, 60, m, 7, m)
        rng = scenes.SplitMix64(99 + m)
        u = rng.uniform(3 * m).reshape(m, 3)
        sc["obstacles"][:m, 0:3] = sc["start"] + 0.12 + 0.1 * u          # a cluster 0.2-0.35 m from the start
        sc["obstacles"][:m, 6] = 0.02
        sc["detect_shell_rad"] = 0.6
        hip, _ = run_both(pmaf, oracle, scenes, sc, 3)
        hip.close()


def test_blocking_wait_flag_gives_the_same_results(pmaf, oracle, scenes):
    """PMAF_FLAG_BLOCKING_WAIT: pmaf_tick sleeps on the manager kernel's completion
    event instead of spinning on the mailbox -- same results, and the host is
    mostly idle while ticks are issued back to back"""
    import time
    sc = scenes.config_scene("C2", scene_id=4)
    hip, ora = make_pair(pmaf, oracle, sc, blocking_wait=True)
    t0, c0 = time.perf_counter(), time.process_time()
    best = [hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for t in range(150)]
    hip.stop()
    wall, cpu = time.perf_counter() - t0, time.process_time() - c0
    assert best == [ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for t in range(150)]
    assert_state_equal(hip, ora)
    assert cpu < 0.7 * wall   # the spinning default burns a full core: cpu ~ wall
    print("blocking wait: %d ticks in %.1f ms wall, %.1f ms CPU" % (150, wall * 1e3, cpu * 1e3))
    hip.close()


@pytest.mark.parametrize("one_wave", [False, True])
@pytest.mark.parametrize("m_field", [70, 128, 200])
def test_closest_other_table_life_cycle(pmaf, oracle, scenes, m_field, monkeypatch, one_wave):
    """DevView::closest_idx (multi-slot wave-per-agent kernels): k_manager computes the Obstacle / GoalObstacle
    heuristics' closest-other answers once per NEW obstacle list at rest, and the rollouts' first-contact latches read
    them instead of scanning (B/src/cf_agent.cpp:434-446 / :480-492). Every agent an Obstacle or GoalObstacle one; the
    list stays, moves to other rest positions, starts moving (no table: the scan), comes to rest again, and a handle
    restored from a checkpoint carries on -- all bit-exact against the oracle, rotation vectors included."""
    if one_wave:   # the one-wave kernels (2 / 4 obstacle slots per lane) instead of k_rollout_mw
        monkeypatch.setenv("PMAF_MW", "0")
    N, H = 10, 90
    types = np.array([2, 3] * (N // 2), dtype=np.int32)
    sc = scenes.synthetic_scene(N, H, m_field, 6, 123 + m_field, agent_types=types)
    # a ring of obstacles around the start-goal line, two of them at EQUAL distance from a third (index ties)
    k = min(m_field, 40)
    ang = np.linspace(0.0, 2 * np.pi, k, endpoint=False)
    sc["obstacles"][:k, 0] = np.linspace(-0.45, 0.45, k)
    sc["obstacles"][:k, 1] = 0.12 * np.cos(ang * 3)
    sc["obstacles"][:k, 2] = 0.7 + 0.12 * np.sin(ang * 3)
    sc["obstacles"][1, :3] = sc["obstacles"][0, :3] + [0.0, 0.05, 0.0]
    sc["obstacles"][2, :3] = sc["obstacles"][0, :3] - [0.0, 0.05, 0.0]
    hip, ora = make_pair(pmaf, oracle, sc, lanes_per_agent=64)
    args = (sc["dt"], sc["cost_gains"], sc["ws_limits"])
    rng = np.random.default_rng(m_field)

    cur = [None]

    def both(obs, n):
        if obs is not None:
            cur[0] = obs
        for _ in range(n):
            assert hip.tick(obs, *args) == ora.tick(cur[0], *args)   # (None: the planner keeps its live list)
        hip.stop()
        assert_state_equal(hip, ora)
        assert hip.known().sum() > 0
        np.testing.assert_array_equal(np.ascontiguousarray(hip.rot_vecs()).view(np.uint64),
                                      np.ascontiguousarray(ora.rot_vecs()).view(np.uint64))

    obs = sc["obstacles"].copy()
    both(obs, 3)                    # table computed at the first reset, reused by the next rollouts (same list)
    both(None, 2)                   # no list handed over: the live one (and the table) stay
    obs2 = obs.copy()
    obs2[:m_field, :3] += rng.uniform(-0.03, 0.03, (m_field, 3))
    both(obs2, 2)                   # other rest positions: recomputed
    obs3 = obs2.copy()
    obs3[5, 3:6] = [0.0, 0.02, 0.0]   # one obstacle moves: no table, the latches scan
    both(obs3, 2)
    both(obs2, 2)                   # at rest again
    obs4 = obs2.copy()
    obs4[-1, :3] = [0.1, 0.3, 0.7]   # only the trailing (repulsive) obstacle differs: not a field obstacle, the table stays
    both(obs4, 2)
    blob = hip.save_state()
    hip2 = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"], lanes_per_agent=64)
    hip2.set_initial_position(sc["start"])
    hip2.load_state(blob)
    hip.close()
    hip = hip2
    both(None, 2)                   # restored handle: table from the blob, recomputed at its first reset
    both(obs, 2)
    hip.close()


def test_tick_times_on_the_library_clock(pmaf, scenes):
    """pmaf_get_tick_times_us: one (enqueue, set-point) pair per pmaf_tick, oldest first, enqueue <= set-point, a
    set-point within a millisecond of the call on an idle stream, and the record is cleared by the read"""
    sc = scenes.config_scene("C2", scene_id=2)
    h = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    h.set_initial_position(sc["start"])
    for k in range(7):
        h.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        h.stop()
    enq, sp = h.tick_times_us()
    assert enq.shape == (7,) and sp.shape == (7,)
    assert np.all(enq > 0) and np.all(enq <= sp) and np.all(sp[1:] < 1000.0)
    e2, s2 = h.tick_times_us()
    assert e2.size == 0 and s2.size == 0
    h.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    h.stop()
    assert h.tick_times_us(max_n=4)[1].shape == (1,)
    print("set-point on the host %.1f us after the call (median of 6), both launches enqueued after %.1f us"
          % (float(np.median(sp[1:])), float(np.median(enq[1:]))))
    h.close()


@pytest.mark.parametrize("config,ticks", [("C1", 10), ("C2", 8), ("C3", 2)])
@pytest.mark.parametrize("case", ["mass", "k_attr_zero_some", "forced_general"])
def test_w64_general_step_mass_and_zero_attractor_gain(pmaf, oracle, scenes, monkeypatch, config, ticks, case):
    """round 3: the wave-per-agent kernels exist with a PLAIN step (every k_attr != 0, unit mass -- decided by
    pmaf_create) and with the general one (attractorForce's `k_attr != 0` test B/src/cf_agent.cpp:184, the division by
    the mass :254). Non-unit mass, a population in which SOME agents have k_attr == 0, and the general step forced onto
    a plain population must all match the oracle bit for bit on the short-list, one-slot and two-slot kernels."""
    sc = dict(scenes.config_scene(config))
    if case == "mass":
        sc["agent_mass"] = 2.5
    elif case == "k_attr_zero_some":
        ka = np.full(int(sc["n_agents"]), float(np.asarray(sc["k_attr"]).reshape(-1)[0]))
        ka[1::3] = 0.0
        sc["k_attr"] = ka
    else:
        monkeypatch.setenv("PMAF_PLAIN_STEP", "0")
    hip, _ = run_both(pmaf, oracle, scenes, sc, ticks)
    assert hip.launch_config()["lanes_per_agent"] == 64
    hip.close()


@pytest.mark.parametrize("blob_case", ["mass", "k_attr_zero_some"])
def test_checkpoint_carries_the_step_variant(pmaf, oracle, scenes, blob_case):
    """round 3: pmaf_create picks the wave-per-agent kernels' PLAIN step from the gains and the mass; a state blob brings
    its OWN gains and mass, so pmaf_load_state must pick again -- a blob saved from a population with non-unit mass (or
    some k_attr == 0) restored into a handle created with the defaults continues like the oracle of the blob's scene"""
    sc = dict(scenes.config_scene("C2", scene_id=1))
    if blob_case == "mass":
        sc["agent_mass"] = 2.5
    else:
        ka = np.full(int(sc["n_agents"]), float(np.asarray(sc["k_attr"]).reshape(-1)[0]))
        ka[::4] = 0.0
        sc["k_attr"] = ka
    hip, ora = make_pair(pmaf, oracle, sc)
    obs = sc["obstacles"]
    for t in range(6):
        assert hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]) == ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    blob = hip.save_state()
    hip.stop(); hip.close()
    plain = scenes.config_scene("C2", scene_id=1)          # unit mass, k_attr != 0 everywhere: the PLAIN step
    hip2 = pmaf.PmafPlanner(plain, device=0)
    hip2.load_state(blob)
    for t in range(6):
        assert hip2.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]) == ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip2.stop()
    assert_state_equal(hip2, ora)
    np.testing.assert_allclose(hip2.costs(), ora.costs(), rtol=0, atol=0)
    hip2.close()

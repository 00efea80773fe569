"""k_rollout_mw (csrc/pmaf_k_mw.hip): W waves per agent for 62..256 field obstacles -- every wave the one-slot step on its
own <= 64 obstacles, one LDS hand-off per step, the ordered force sum over the waves' lists in ascending obstacle index.
Bit-exact against the oracle like the one-wave kernels (which PMAF_MW=0 keeps: tests/test_parity_gpu.py runs its
multi-obstacle cases through both)."""
import numpy as np
import pytest

from test_parity_gpu import _portable_exp_oracle, make_pair, assert_state_equal, run_both  # noqa: F401

pytestmark = pytest.mark.gpu


def _expected_split(m):
    waves = max(2, (m + 63) // 64)
    return waves, (m + waves - 1) // waves


@pytest.mark.parametrize("dynamic", [False, True])
@pytest.mark.parametrize("m", [61, 62, 64, 100, 122, 123, 128, 129, 183, 184, 192, 200, 244, 245, 256])
def test_every_split_matches_the_oracle(pmaf, oracle, scenes, m, dynamic):
    """obstacle counts on both sides of every boundary of the split: 2 / 3 / 4 waves, <= 61 obstacles per wave (riders in
    lanes 61..63, the sweep's norms in the tail's sequence) and 62..64 (the sweep takes its own); every heuristic type
    in the population; obstacles at rest (closest-other table) and moving (mirror + search)"""
    sc = scenes.synthetic_scene(14, 110, m, 7, m, dynamic=dynamic)
    hip, _ = run_both(pmaf, oracle, scenes, sc, 3, dynamic=dynamic)
    cfg = hip.launch_config()
    assert (cfg["waves_per_agent"], cfg["obstacles_per_wave"]) == _expected_split(m)
    hip.close()


@pytest.mark.parametrize("waves", [3, 4])
def test_more_waves_than_needed(pmaf, oracle, scenes, monkeypatch, waves):
    """PMAF_MW=3|4: BASELINE C3's 128 obstacles on three / four waves (43 / 32 per wave)"""
    monkeypatch.setenv("PMAF_MW", str(waves))
    sc = scenes.config_scene("C3")
    hip, _ = run_both(pmaf, oracle, scenes, sc, 2)
    assert hip.launch_config()["waves_per_agent"] == waves
    hip.close()


def test_one_wave_kernels_on_request_and_beyond_one_wave_per_simd(pmaf, oracle, scenes, monkeypatch):
    """PMAF_MW=0 keeps the one-wave kernels; so does a launch that could not give every block a CU of its own
    (more than 256 agents: round 5's rule, profiles/r5_mw_rule_sweep.txt)"""
    sc = scenes.synthetic_scene(256, 20, 128, 7, 3)
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    assert hip.launch_config()["waves_per_agent"] == 2     # a CU per block: the split kernel
    hip.close()
    sc = scenes.synthetic_scene(300, 40, 128, 7, 3)
    hip, ora = make_pair(pmaf, oracle, sc)
    assert hip.launch_config()["waves_per_agent"] == 1
    for _ in range(2):
        assert hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"]) == \
            ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    assert_state_equal(hip, ora)
    hip.close()
    monkeypatch.setenv("PMAF_MW", "0")
    sc = scenes.synthetic_scene(14, 60, 128, 7, 4)
    hip, _ = run_both(pmaf, oracle, scenes, sc, 2)
    assert hip.launch_config()["waves_per_agent"] == 1
    hip.close()


def test_signed_zeros_gate_runs_and_early_stops(pmaf, oracle, scenes):
    """-0.0 coordinates / velocity components of obstacles at rest (the mirror's second buffer holds the positions
    after ONE predictObstacles), a start inside the gate for many steps (no exchange while it is closed: the buffers'
    parity follows the exchanges, not the steps) and agents that reach the goal before the horizon ends"""
    sc = scenes.synthetic_scene(12, 400, 70, 9, 7)
    obs = sc["obstacles"]
    obs[0, :3] = [0.0, -0.0, 0.7]
    obs[1, :3] = [-0.0, 0.05, 0.7]
    obs[1, 3:6] = [-0.0, 0.0, -0.0]
    obs[2, :3] = [0.2, -0.0, 0.7]
    obs[2, 3:6] = [0.0, -0.0, 0.0]
    obs[65, :3] = [0.1, -0.0, 0.75]      # second wave
    obs[65, 3:6] = [0.0, 0.0, -0.0]
    sc["goal"] = np.array([0.05, 0.0, 0.7])   # close enough for early stops within the horizon
    hip, _ = run_both(pmaf, oracle, scenes, sc, 4)
    assert hip.launch_config()["waves_per_agent"] == 2
    assert (hip.paths()[1] < 401).any()
    hip.close()


def test_contracted_policy_within_tolerance(pmaf, oracle, scenes):
    """PMAF_FLAG_CONTRACTED on the split kernel: selected trajectory within the north star's 1e-5 m of the libm oracle
    over closed-loop ticks at C3, same best-index sequence (tests/test_tolerance_gpu.py holds the long runs)"""
    oracle.set_exp_mode(0)
    sc = scenes.config_scene("C3")
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"], contracted=True)
    ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"]); ora.set_initial_position(sc["start"])
    assert hip.launch_config()["waves_per_agent"] == 2
    for _ in range(6):
        bh = hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        bo = ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        assert bh == bo
        hip.stop()
        (ph, nh), (po, no) = hip.paths(), ora.paths()
        assert nh[bo] == no[bo]
        assert np.abs(ph[bo, :no[bo]] - po[bo, :no[bo]]).max() <= 1e-5
    hip.close()


def test_batched_populations_on_split_blocks(pmaf, oracle, scenes):
    """P populations in one handle (grid N x P of multi-wave blocks): some at rest (closest-other table per population),
    some moving (mirror on the step's parity), against one oracle per population"""
    scs = [scenes.synthetic_scene(20, 120, 100, 5, 40 + sid, dynamic=(sid % 2 == 1)) for sid in range(4)]
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    assert hip.launch_config()["waves_per_agent"] == 2
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    obs = np.stack([s["obstacles"] for s in scs])
    for t in range(5):
        bh = hip.tick(obs, scs[0]["dt"], scs[0]["cost_gains"], scs[0]["ws_limits"])
        bo = [o.tick(obs[i], scs[0]["dt"], scs[0]["cost_gains"], scs[0]["ws_limits"]) for i, o in enumerate(oras)]
        np.testing.assert_array_equal(bh, bo)
        obs = np.stack([scenes.advance_live_obstacles(o) if i % 2 == 1 else o for i, o in enumerate(obs)])
    hip.stop()
    ph, nh = hip.paths()
    for i, o in enumerate(oras):
        po, no = o.paths()
        np.testing.assert_array_equal(nh[i], no)
        assert np.array_equal(ph[i], po, equal_nan=True)
        np.testing.assert_array_equal(hip.rot_vecs()[i], o.rot_vecs())
        np.testing.assert_array_equal(hip.known()[i], o.known())
    hip.close()


@pytest.mark.parametrize("cfg", ["C3", "M200_dynamic"])
def test_split_and_one_wave_kernels_agree_over_a_long_closed_loop(pmaf, scenes, monkeypatch, cfg):
    """size-independent check at full size: 150 closed-loop ticks (episode restarts every 50, as the bench does) through
    k_rollout_mw and, with PMAF_MW=0, through the one-wave kernels -- identical best-index sequences, set-points, paths,
    rotation vectors and costs, bit for bit (the oracle comparison at this size runs for a few ticks only)"""
    if cfg == "C3":
        sc, dyn = scenes.config_scene("C3"), False
    else:
        sc, dyn = scenes.synthetic_scene(96, 300, 200, 8, 21, dynamic=True), True

    def run(env):
        if env is None:
            monkeypatch.delenv("PMAF_MW", raising=False)
        else:
            monkeypatch.setenv("PMAF_MW", env)
        hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
        hip.set_initial_position(sc["start"])
        waves = hip.launch_config()["waves_per_agent"]
        obs = sc["obstacles"].copy()
        best, setp = [], []
        for t in range(150):
            if t % 50 == 0 and t:
                hip.set_initial_position(sc["start"])
                obs = sc["obstacles"].copy()
            best.append(hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]))
            setp.append(np.concatenate([hip.last_next_pos.ravel(), hip.last_next_vel.ravel()]))
            if dyn:
                obs = scenes.advance_live_obstacles(obs)
        hip.stop()
        out = (waves, np.array(best), np.array(setp), hip.paths(), hip.costs(), hip.rot_vecs(), hip.known(), hip.min_obs_dist())
        hip.close()
        return out

    a, b = run(None), run("0")
    assert a[0] >= 2 and b[0] == 1
    np.testing.assert_array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2], equal_nan=True)
    assert np.array_equal(a[3][0], b[3][0], equal_nan=True) and np.array_equal(a[3][1], b[3][1])
    for x, y in zip(a[4:], b[4:]):
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)

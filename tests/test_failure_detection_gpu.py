"""Failure detection on the tick and the selected trajectory in pinned memory (ABI 5, include/pmaf.h), through the
C-ABI on the GPU:
  * health word of a tick (NaN set-point / force, acceleration clamp) with the Had heuristic's degenerate geometry
    driving the REAL agent (B/src/cf_agent.cpp:599-611; the reference's consumer only logs the NaN,
    B/src/costp_controller.cpp:317-319);
  * pmaf_tick's wall-clock bound (PMAF_TICK_TIMEOUT_S) with the sequence number withheld by the debug hook;
  * pmaf_view_winner_path against paths()[best];
  * the stepping API's start rule and the closest-other table under an external rollout kernel."""
import os
import subprocess
import time

import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _had_on_the_goal_line(scenes, n_agents=4, horizon=60):
    """every agent a Had heuristic, ONE field obstacle whose centre lies exactly on the start-goal line"""
    sc = scenes.synthetic_scene(n_agents, horizon, 1, 9, 2)
    sc["start"] = np.array([-0.6, 0.0, 0.7])
    sc["goal"] = np.array([0.6, 0.0, 0.7])
    sc["obstacles"][0] = [0.0, 0.0, 0.7, 0, 0, 0, 0.05]
    sc["agent_types"] = np.full(n_agents, 6, dtype=np.int32)   # PMAF_HAD_HEURISTIC
    return sc


def test_health_word_reports_the_real_agents_nan(pmaf, oracle, scenes):
    oracle.set_exp_mode(1)
    try:
        sc = _had_on_the_goal_line(scenes)
        hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
        ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
        hip.set_initial_position(sc["start"])
        ora.set_initial_position(sc["start"])
        first_nan = None
        for t in range(400):
            bh = hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])   # returns PMAF_OK with the NaN
            bo = ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
            assert bh == bo
            rh, ro = hip.real_state(), ora.real_state()
            for a, b in zip(rh, ro):   # the NaN pattern is the reference's (oracle), bit for bit elsewhere
                np.testing.assert_array_equal(a, b)
            hb = int(hip.health())
            nan_now = bool(np.isnan(ro[0]).any())
            assert bool(hb & hip.HEALTH_SETPOINT_NAN) == nan_now, (t, hb)
            if nan_now:
                assert hb & hip.HEALTH_FORCE_NAN
                first_nan = t
                break
            assert not (hb & hip.HEALTH_FORCE_NAN)
        assert first_nan is not None and first_nan > 50, "the real agent never reached the degenerate obstacle"
        print("real agent's set-point NaN at tick %d, health word %d" % (first_nan, hb))
        hip.close()
    finally:
        oracle.set_exp_mode(0)


def test_health_word_reports_the_acceleration_clamp(pmaf, scenes):
    """a repulsive obstacle almost touching the real agent: |a| > 13 -> clamped (B/src/cf_agent.cpp:255-257)"""
    sc = scenes.synthetic_scene(6, 40, 4, 9, 5)
    sc["obstacles"][-1] = [sc["start"][0] + 0.151, sc["start"][1], sc["start"][2], 0, 0, 0, 0.1]   # surface 1 mm away
    sc["k_repel"] = 5.0
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    assert int(hip.health()) & hip.HEALTH_ACC_CLAMPED
    assert not int(hip.health()) & (hip.HEALTH_SETPOINT_NAN | hip.HEALTH_FORCE_NAN)
    far = sc["obstacles"].copy()
    far[-1, :3] = 100.0
    hip.tick(far, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.tick(far, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    assert int(hip.health()) == 0
    hip.close()


def test_tick_time_limit_when_the_sequence_number_never_arrives(pmaf, scenes, monkeypatch):
    """fault injection: the manager kernel does not publish its sequence number while a long rollout keeps the stream
    busy -> pmaf_tick must give up after PMAF_TICK_TIMEOUT_S with PMAF_ERR_DEVICE instead of spinning for ever; the
    handle works again afterwards"""
    monkeypatch.setenv("PMAF_TICK_TIMEOUT_S", "0.002")
    sc = scenes.synthetic_scene(512, 6000, 128, 3, 1)       # ~15 ms of rollout per tick
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    hip.debug_withhold_mailbox(True)
    t0 = time.perf_counter()
    with pytest.raises(pmaf.PmafError) as ei:
        hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    waited = time.perf_counter() - t0
    assert ei.value.code == -2 and "time limit" in str(ei.value), str(ei.value)
    assert waited < 0.012, "gave up only after %.1f ms (limit 2 ms, rollout ~15 ms)" % (waited * 1e3)
    hip.debug_withhold_mailbox(False)
    # the abandoned tick's kernels are still queued (they read the call's staging buffers and write the mailbox later):
    # no further tick until the stream has been drained
    with pytest.raises(pmaf.PmafError) as ei2:
        hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    assert ei2.value.code == -3 and "pmaf_stop" in str(ei2.value), str(ei2.value)
    hip.stop()
    b = hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    assert 0 <= b < 512
    hip.close()


@pytest.mark.parametrize("P", [1, 3])
def test_winner_path_in_pinned_memory_is_the_scored_path_of_the_best_agent(pmaf, scenes, P):
    scs = [scenes.synthetic_scene(40, 120, 20, 5, s) for s in range(P)]
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs if P > 1 else scs[0], device=0, mgr_init_pos=starts if P > 1 else starts[0])
    hip.set_initial_position(starts if P > 1 else starts[0])
    obs = np.stack([s["obstacles"] for s in scs])
    sc = scs[0]
    with pytest.raises(pmaf.PmafError):
        hip.winner_path()                       # not enabled
    hip.enable_winner_path()
    for t in range(25):
        hip.stop()
        paths, n = hip.paths()
        paths, n = (paths[None], n[None]) if P == 1 else (paths, n)
        b = np.atleast_1d(hip.tick(obs if P > 1 else obs[0], sc["dt"], sc["cost_gains"], sc["ws_limits"]))
        wp, wn, wa = hip.winner_path()
        wp, wn, wa = ([wp], [wn], [wa]) if P == 1 else (wp, wn, wa)
        for p in range(P):
            assert wa[p] == b[p] and wn[p] == n[p, b[p]]
            np.testing.assert_array_equal(wp[p], paths[p, b[p], :n[p, b[p]]])
    # a selection that is not a tick publishes its path too
    hip.stop()
    paths, n = hip.paths()
    paths, n = (paths[None], n[None]) if P == 1 else (paths, n)
    b = np.atleast_1d(hip.evaluate(sc["cost_gains"], sc["ws_limits"]))
    wp, wn, wa = hip.winner_path()
    wp, wn, wa = ([wp], [wn], [wa]) if P == 1 else (wp, wn, wa)
    for p in range(P):
        assert wa[p] == b[p]
        np.testing.assert_array_equal(wp[p], paths[p, b[p], :n[p, b[p]]])
    us = hip.winner_path_times_us()
    assert len(us) == 25 and (us > 0).all()
    print("tick entry -> winner path on the host: median %.1f us (back-to-back ticks: includes the previous rollout)" % np.median(us))
    hip.enable_winner_path(False)
    hip.tick(obs if P > 1 else obs[0], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.close()


def test_facade_plantick_reports_a_nan_setpoint_throws_on_opt_in_and_serves_the_selected_path(hip_lib):
    """default = the reference's behaviour (NaN published, health word set, no throw: ADVICE r4); opt-in throw at the same tick"""
    exe = conftest.exe(os.path.join(ROOT, "tests", "cpp", "facade_tick"))
    r = subprocess.run([exe, "health"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode().split()
    assert out[0] == "H" and int(out[1]) > 50 and int(out[2]) & 3 == 3
    assert out[3] == "S" and int(out[4]) == int(out[1])     # every tick before the throw checked its selected path


def test_start_after_set_agent_positions_needs_a_reset(pmaf, oracle, scenes):
    """pmaf_set_agent_positions / pmaf_set_agent_pos_and_vels put the handle into the `stepped` state --
    pmaf_start then fails with PMAF_ERR_STATE (documented deviation) until a reset; the stepping calls
    themselves keep working and match the oracle"""
    sc = scenes.static1_scene(8, 60)
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    for setter in (lambda: hip.set_agent_positions(sc["start"] + 0.01),
                   lambda: hip.set_agent_pos_and_vels(sc["start"] + 0.02, [0.05, 0.0, 0.0])):
        setter()
        with pytest.raises(pmaf.PmafError) as ei:
            hip.start()
        assert ei.value.code == -3 and "pmaf_reset_agents" in str(ei.value)
        hip.move_agents(sc["obstacles"], sc["dt"], 3)          # the stepping API itself is fine
        with pytest.raises(pmaf.PmafError):
            hip.start()
        pos, vel, _ = hip.real_state()
        hip.reset_agents(pos, vel, sc["obstacles"])
        hip.start()                                            # legal again
        hip.stop()
        assert (hip.n_points() > 1).any()
    hip.close()


def test_sampled_kernel_timing_and_resident_obstacle_lists(pmaf, oracle, scenes):
    """pmaf_set_profiling(n): every n-th rollout launch carries HIP events; pmaf_get_launch_count counts them all.
    And pmaf_tick with the SAME obstacle list as the resident one (what the reference's node passes every tick) must
    behave exactly like a tick that hands a list over -- bit-exact against the oracle across same / changed / same lists"""
    oracle.set_exp_mode(1)
    try:
        sc = scenes.config_scene("C1")
        hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
        ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
        hip.set_initial_position(sc["start"])
        ora.set_initial_position(sc["start"])
        hip.set_profiling(4)
        hip.reset_kernel_stats()
        obs = sc["obstacles"].copy()
        for t in range(40):
            if t in (10, 11, 25):                      # a changed list, changed again, and back to an earlier one
                obs = obs.copy()
                obs[2, :3] += 0.01 if t != 25 else -0.02
            bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
            bo = ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
            assert bh == bo
            np.testing.assert_array_equal(hip.real_state()[0], ora.real_state()[0])
        hip.stop()
        ph, nh = hip.paths()
        po, no = ora.paths()
        np.testing.assert_array_equal(nh, no)
        np.testing.assert_array_equal(ph, po)
        ms, timed, steps = hip.kernel_stats()
        assert hip.launch_count() == 40 and timed == 10 and ms > 0
        assert steps == int(sum(no - 1)) or steps > 0
        hip.close()
    finally:
        oracle.set_exp_mode(0)

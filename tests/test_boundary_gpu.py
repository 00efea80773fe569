"""Boundary calls of the planner node that the open-loop single-goal tick never takes (missing 4 / 5),
HIP through the C-ABI against the CPU oracle at tolerance 0:

  * closed loop -- `if (!open_loop_) setRealEEAgentPosition(p)` in front of every tick
    (B/src/panda_bimanual_control.cpp:333-335 -> B/src/cf_manager.cpp:216-218 -> RealCfAgent::setPosition = push_back,
    B/src/cf_agent.cpp:44-46): the measured position trails the set-point by a deterministic tracking error;
    getPlannedTrajectory() holds the extra point of every tick;
  * prediction_freq_multiple != 1 -- the rollouts integrate with mult * dt (B/src/cf_manager.cpp:118-123) while the real
    agent's step keeps dt (B/src/panda_bimanual_control.cpp:348);
  * a new goal -- re-init with the best agent (id for the 0.9 hysteresis, heuristic type, ITS Random vectors) surviving
    (B/src/cf_manager.cpp:344-354), also mid-run while a Random agent leads;
  * the same three through the C++ facade + planner-node mirror (tools/plan_task on task files in the reference's
    schema: tests/golden/tasks/{static1_closed_loop,dyn1_freq2,dyn1_two_goals}.yaml)."""
import copy
import os
import subprocess

import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu

ROOT = conftest.ROOT
EXE = conftest.exe(os.path.join(ROOT, "tools", "plan_task"))
TASKS = os.path.join(ROOT, "tests", "golden", "tasks")
LAG = 0.3   # share of the last step the controller has NOT covered when it reports its position


@pytest.fixture(autouse=True)
def _portable_exp_oracle(oracle):
    oracle.set_exp_mode(1)
    yield
    oracle.set_exp_mode(0)


def _same(a, b):
    np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def _assert_all_equal(hip, ora):
    hip.stop()
    ph, nh = hip.paths()
    po, no = ora.paths()
    _same(nh, no)
    _same(ph, po)
    for f in ("costs", "min_obs_dist", "path_lengths", "agent_vel", "success", "known", "rot_vecs", "real_path"):
        _same(getattr(hip, f)(), getattr(ora, f)())
    for a, b in zip(hip.real_state(), ora.real_state()):
        _same(a, b)
    for a, b in zip(hip.real_known(), ora.real_known()):
        _same(a, b)
    assert hip.best_type() == ora.best_type() and hip.best_id() == ora.best_id()
    assert hip.dist_from_goal() == ora.dist_from_goal()


def _hip_tick(hip, sc, obs, dt, style):
    """one planCallback: the fused pmaf_tick or the node's five calls"""
    if style == "tick":
        return hip.tick(obs, dt, sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    b = hip.evaluate(sc["cost_gains"], sc["ws_limits"])
    hip.move_real(obs, dt, 1, b)
    pos, vel, _ = hip.real_state()
    hip.reset_agents(pos, vel, obs)
    hip.start()
    return b


def _closed_loop(pmaf, oracle, scenes, sc, n_ticks, style, dynamic, until_reached=False):
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    ora.set_initial_position(sc["start"])
    obs = sc["obstacles"].copy()
    measured = np.asarray(sc["start"], dtype=np.float64).copy()
    for t in range(n_ticks):
        hip.set_real_position(measured)          # setRealEEAgentPosition(p), :333-335
        ora.set_real_position(measured)
        _same(hip.real_state()[0], measured)     # getNextPosition() is the measured position until the step
        bh = _hip_tick(hip, sc, obs, sc["dt"], style)
        bo = ora.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        assert bh == bo, (t, bh, bo)
        sp = np.asarray(ora.real_state()[0])
        _same(hip.real_state()[0], sp)
        measured = sp - LAG * (sp - measured)    # the controller trails the set-point
        if dynamic:
            obs = scenes.advance_live_obstacles(obs)
        if until_reached and ora.dist_from_goal() < 0.01:
            break
    _assert_all_equal(hip, ora)
    # RealCfAgent::setPosition pushes: the planned trajectory holds TWO points per tick here (measured + stepped)
    assert len(hip.real_path()) == 2 + 2 * (t + 1)
    hip.close()
    return t + 1


@pytest.mark.parametrize("style", ["tick", "five_calls"])
@pytest.mark.parametrize("cfg", ["C1", "C2"])
def test_closed_loop_measured_position_before_every_tick(pmaf, oracle, scenes, cfg, style):
    sc = scenes.config_scene(cfg)
    _closed_loop(pmaf, oracle, scenes, sc, 40 if cfg == "C1" else 15, style, False)


def test_closed_loop_c2_moving_obstacles(pmaf, oracle, scenes):
    sc = scenes.config_scene("C2", scene_id=3, dynamic=True)
    _closed_loop(pmaf, oracle, scenes, sc, 15, "tick", True)


@pytest.mark.parametrize("style", ["tick", "five_calls"])
def test_closed_loop_dyn1_until_reached(pmaf, oracle, scenes, style):
    """the dual_arms_dyn1 scene to `reached` with the tracking error in the loop (the lag makes it longer than the
    open-loop run's 745 ticks)"""
    sc = scenes.dyn1_scene(10, 400)
    n = _closed_loop(pmaf, oracle, scenes, sc, 2500, style, True, until_reached=True)
    assert 745 < n < 2500, n


@pytest.mark.parametrize("P", [2, 6])
def test_closed_loop_batched_populations(pmaf, oracle, scenes, P):
    """P populations in one handle, each with its own measured position: up to four populations the positions travel by
    value in the manager kernel's arguments, beyond that through the pinned staging buffer (ManagerArgs)"""
    scs = [scenes.synthetic_scene(24, 80, 20, 5, 60 + sid, dynamic=(sid % 2 == 1)) for sid in range(P)]
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    obs = np.stack([s["obstacles"] for s in scs])
    measured = starts.copy()
    sc = scs[0]
    for t in range(12):
        hip.set_real_position(measured)
        for p, o in enumerate(oras):
            o.set_real_position(measured[p])
        if t % 2:
            bh = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        else:   # the individual calls: the position is consumed by the evaluate (a launch without a step)
            hip.stop()
            bh = hip.evaluate(sc["cost_gains"], sc["ws_limits"])
            hip.move_real(obs, sc["dt"], 1, bh)
            pos, vel, _ = hip.real_state()
            hip.reset_agents(pos, vel, obs)
            hip.start()
        bo = [o.tick(obs[p], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for p, o in enumerate(oras)]
        _same(bh, bo)
        sp = np.stack([o.real_state()[0] for o in oras])
        _same(hip.real_state()[0], sp)
        measured = sp - LAG * (sp - measured)
        obs = np.stack([scenes.advance_live_obstacles(o) if p % 2 == 1 else o for p, o in enumerate(obs)])
    hip.stop()
    ph, nh = hip.paths()
    for p, o in enumerate(oras):
        po, no = o.paths()
        _same(nh[p], no)
        _same(ph[p], po)
        _same(hip.real_path(p), o.real_path())
    hip.close()


def test_closed_loop_position_survives_evaluate_and_checkpoint(pmaf, oracle, scenes):
    """the measured position is handed to the NEXT manager launch through pinned memory: an evaluate alone (no step),
    a state blob taken in between and the winner-record packing must all see it"""
    sc = scenes.config_scene("C1")
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    ora.set_initial_position(sc["start"])
    for t in range(3):
        hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    m = np.asarray(ora.real_state()[0]) + np.array([0.002, -0.001, 0.0015])
    hip.set_real_position(m)
    ora.set_real_position(m)
    blob = hip.save_state()                       # (flushes the pending position into the device state)
    hip2 = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip2.load_state(blob)
    for h in (hip, hip2):
        h.stop()
        assert h.evaluate(sc["cost_gains"], sc["ws_limits"]) == ora.evaluate(sc["cost_gains"], sc["ws_limits"])
        _same(h.real_state()[0], m)
    m2 = m + np.array([-0.0005, 0.0, 0.001])
    hip.set_real_position(m2)                     # consumed by an evaluate (a manager launch without a step) ...
    hip2.set_real_position(m2)
    ora.set_real_position(m2)
    for h in (hip, hip2):
        b = h.evaluate(sc["cost_gains"], sc["ws_limits"])
        h.move_real(sc["obstacles"], sc["dt"], 1, b)   # ... and the step that follows starts from it
    bo = ora.evaluate(sc["cost_gains"], sc["ws_limits"])
    ora.move_real(sc["obstacles"], sc["dt"], 1, bo)
    for h in (hip, hip2):
        for a, b_ in zip(h.real_state(), ora.real_state()):
            _same(a, b_)
        _same(h.real_path(), ora.real_path())
        h.close()


@pytest.mark.parametrize("mult", [2, 3])
@pytest.mark.parametrize("case", ["C2", "C2dyn", "dyn1"])
def test_prediction_freq_multiple(pmaf, oracle, scenes, case, mult):
    """rollout dt = mult * dt (also in the agents' predictObstacles), real step dt"""
    if case == "dyn1":
        sc = scenes.dyn1_scene(10, 300)
    else:
        sc = scenes.config_scene("C2", scene_id=3 if case == "C2dyn" else 0, dynamic=case == "C2dyn")
    dt_real = sc["dt"]
    sc = copy.deepcopy(sc)
    sc["dt"] = mult * dt_real                    # CfManager::init: prediction_freq_multiple * delta_t
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    ora.set_initial_position(sc["start"])
    obs = sc["obstacles"].copy()
    for t in range(60 if case == "dyn1" else 12):
        style = "tick" if t % 3 else "five_calls"
        bh = _hip_tick(hip, sc, obs, dt_real, style)
        bo = ora.tick(obs, dt_real, sc["cost_gains"], sc["ws_limits"])
        assert bh == bo, (t, bh, bo)
        if case != "C2":
            obs = scenes.advance_live_obstacles(obs)
    _assert_all_equal(hip, ora)
    # the multiple is really in the rollouts: the same scene at multiple 1 predicts other paths
    sc1 = copy.deepcopy(sc)
    sc1["dt"] = dt_real
    ref = oracle.OraclePlanner(sc1, mgr_init_pos=sc["start"])
    ref.set_initial_position(sc["start"])
    ref.tick(sc["obstacles"], dt_real, sc["cost_gains"], sc["ws_limits"])
    ref.tick(sc["obstacles"], dt_real, sc["cost_gains"], sc["ws_limits"])
    assert not np.array_equal(ref.paths()[0], ora.paths()[0])
    hip.close()


def _reinit(make, old, old_sc, new_sc, position):
    """taskCallback's PLAN branch (B/src/panda_bimanual_control.cpp:494-511) on either planner class: the position
    message that arrives while planning is inactive (:364-367), getNextPosition(), init() -- a NEW population at
    CfManager::init_pos_, best_agent_ surviving -- and setInitialPosition(current position)"""
    old.set_initial_position(position)
    cur = np.asarray(old.real_state()[0]).copy()
    bid, btype = old.best_id(), old.best_type()
    new = make(new_sc, position)
    if bid > 0:
        new.set_best(bid, btype, old_sc["random_vecs"][bid - 1])
    new.set_initial_position(cur)
    return new


@pytest.mark.parametrize("cut", [300, None])
def test_new_goal_carries_the_best_agent_over(pmaf, oracle, scenes, cut):
    """dyn1, then a second goal. cut = 300: the goal changes mid-run while Random agent 9 leads (its OLD Random
    vectors drive the real agent until the new population's costs beat 0.9 x the carried index's); None: after
    `reached` (an Obstacle heuristic leads)"""
    sc1 = scenes.dyn1_scene(10, 300)
    mk_h = lambda sc, ip: pmaf.PmafPlanner(sc, device=0, mgr_init_pos=ip)
    mk_o = lambda sc, ip: oracle.OraclePlanner(sc, mgr_init_pos=ip)
    hip, ora = mk_h(sc1, sc1["start"]), mk_o(sc1, sc1["start"])
    hip.set_initial_position(sc1["start"])
    ora.set_initial_position(sc1["start"])
    obs = sc1["obstacles"].copy()
    for t in range(cut or 2000):
        assert hip.tick(obs, sc1["dt"], sc1["cost_gains"], sc1["ws_limits"]) == ora.tick(obs, sc1["dt"], sc1["cost_gains"], sc1["ws_limits"])
        obs = scenes.advance_live_obstacles(obs)
        if cut is None and ora.dist_from_goal() < 0.01:
            break
    _assert_all_equal(hip, ora)
    carried = (ora.best_id(), ora.best_type())
    assert carried == ((9, 5) if cut else (3, 2)), carried      # (id, CfAgent::Type): Random agent 9 / Obstacle heuristic
    sc2 = copy.deepcopy(sc1)
    sc2["goal"] = np.array([-0.45, 0.1, 0.6])
    sc2["obstacles"] = obs.copy()
    sc2["random_vecs"] = scenes.synthetic_scene(10, 300, obs.shape[0] - 1, 77, 5)["random_vecs"]   # init() draws fresh ones
    position = np.asarray(ora.real_state()[0]).copy()
    hip2 = _reinit(mk_h, hip, sc1, sc2, position)
    ora2 = _reinit(mk_o, ora, sc1, sc2, position)
    hip.close()
    seq = []
    for t in range(120):
        bh = hip2.tick(obs, sc2["dt"], sc2["cost_gains"], sc2["ws_limits"])
        bo = ora2.tick(obs, sc2["dt"], sc2["cost_gains"], sc2["ws_limits"])
        assert bh == bo, (t, bh, bo)
        for a, b in zip(hip2.real_state(), ora2.real_state()):
            _same(a, b)
        seq.append(bo)
        obs = scenes.advance_live_obstacles(obs)
    _assert_all_equal(hip2, ora2)
    # the carried agent leads the first ticks of the new goal (equal one-point costs cannot beat 0.9 x its own) ...
    assert seq[0] == carried[0] - 1 and len(set(seq)) > 1, seq[:12]
    # ... and it matters: without the carry the first selection is index 0 and the run differs
    plain = mk_o(sc2, position)
    plain.set_initial_position(position)
    assert plain.tick(sc2["obstacles"], sc2["dt"], sc2["cost_gains"], sc2["ws_limits"]) == 0
    hip2.close()


# ---- the same paths through the C++ facade and the planner-node mirror --------------------------------------------

def _oracle_node_run(oracle, scenes, sc, goals, rv_blocks, max_ticks, goal_ticks=-1, lag=0.0, closed_loop=False, mult=1):
    """tools/plan_task.cpp's loop on the oracle; returns the per-tick rows and the planned-trajectory sizes per goal"""
    start = np.asarray(sc["start"], dtype=np.float64)
    dt_real = sc["dt"]
    obs = sc["obstacles"].copy()
    position = start.copy()
    rows, traj, tick, ora, old_sc = [], [], 0, None, None
    for k, goal in enumerate(goals):
        gsc = copy.deepcopy(sc)
        gsc["goal"] = np.asarray(goal, dtype=np.float64)
        gsc["random_vecs"] = rv_blocks[min(k, len(rv_blocks) - 1)]
        gsc["obstacles"] = obs.copy()
        gsc["dt"] = mult * dt_real
        if ora is None:
            ora = oracle.OraclePlanner(gsc, mgr_init_pos=start)
            ora.set_initial_position(start)
        else:
            ora = _reinit(lambda s, ip: oracle.OraclePlanner(s, mgr_init_pos=ip), ora, old_sc, gsc, position)
        old_sc = gsc
        ip = np.asarray(ora.real_state()[0])
        position = np.array([ip[0], ip[1], (ip[2] + 0.00001) - 0.00001])   # the first published point, echoed (:514-518)
        goal_start = tick
        while tick < max_ticks:
            if closed_loop:
                ora.set_real_position(position)
            b = ora.tick(obs, dt_real, sc["cost_gains"], sc["ws_limits"])
            nxt = np.asarray(ora.real_state()[0]).copy()
            rows.append((tick, b) + tuple(nxt) + (ora.dist_from_goal(),))
            position = nxt - lag * (nxt - position) if lag else nxt
            obs = scenes.advance_live_obstacles(obs)
            tick += 1
            if ora.dist_from_goal() < 0.01:
                break
            if goal_ticks >= 0 and tick - goal_start >= goal_ticks:
                break
        traj.append(len(ora.real_path()))
    return rows, traj


def _plan_task(pmaf, tmp_path, task, sc, rv_blocks, extra):
    rvf = tmp_path / "rv.bin"
    np.ascontiguousarray(np.stack(rv_blocks)).tofile(rvf)
    cmd = [EXE, os.path.join(TASKS, task + ".yaml"), "--start"] + [repr(float(x)) for x in sc["start"]] + \
          ["--random-vecs", str(rvf)] + extra
    r = subprocess.run(cmd, capture_output=True, env=conftest.binary_env(pmaf))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = r.stdout.decode().strip().split("\n")
    data = [l.split() for l in lines if not l.startswith("#")]
    notes = [l for l in lines if l.startswith("# goal")]
    return data, notes


def _compare(data, notes, rows, traj):
    assert len(data) == len(rows)
    for f, ro in zip(data, rows):
        assert int(f[0]) == ro[0] and int(f[1]) == ro[1], (f, ro)
        assert [float(x) for x in f[2:6]] == list(ro[2:6]), (f, ro)
    assert [int(n.split("planned trajectory")[1].split()[0]) for n in notes] == traj


def test_node_closed_loop_task_matches_oracle(pmaf, oracle, scenes, tmp_path, hip_lib):
    sc = scenes.static1_scene(10, 300)
    data, notes = _plan_task(pmaf, tmp_path, "static1_closed_loop", sc, [sc["random_vecs"]], ["--max-ticks", "2500", "--lag", repr(LAG)])
    rows, traj = _oracle_node_run(oracle, scenes, sc, [sc["goal"]], [sc["random_vecs"]], 2500, lag=LAG, closed_loop=True)
    _compare(data, notes, rows, traj)
    assert notes[-1].startswith("# goal reached")
    # and the lag is really in the loop: the open-loop task file takes another course
    d2, _ = _plan_task(pmaf, tmp_path, "static1", sc, [sc["random_vecs"]], ["--max-ticks", "60"])
    assert [x[2:5] for x in d2[:60]] != [x[2:5] for x in data[:60]]


@pytest.mark.parametrize("goal_ticks", [-1, 300])
def test_node_two_goal_task_matches_oracle(pmaf, oracle, scenes, tmp_path, hip_lib, goal_ticks):
    """two plan goals in one task file: the manager is re-initialised towards the second goal from the position the
    first one ended at, the best agent carried over; goal_ticks 300 = the goal changes mid-run (Random agent leading)"""
    sc = scenes.dyn1_scene(10, 300)
    rv2 = scenes.synthetic_scene(10, 300, sc["obstacles"].shape[0] - 1, 77, 5)["random_vecs"]
    goals = [sc["goal"], np.array([-0.45, 0.1, 0.6])]
    extra = ["--max-ticks", "4000"] + (["--goal-ticks", str(goal_ticks)] if goal_ticks >= 0 else [])
    data, notes = _plan_task(pmaf, tmp_path, "dyn1_two_goals", sc, [sc["random_vecs"], rv2], extra)
    rows, traj = _oracle_node_run(oracle, scenes, sc, goals, [sc["random_vecs"], rv2], 4000, goal_ticks=goal_ticks)
    _compare(data, notes, rows, traj)
    assert len(notes) == 2


def test_node_prediction_freq_multiple_task_matches_oracle(pmaf, oracle, scenes, tmp_path, hip_lib):
    sc = scenes.dyn1_scene(10, 300)
    data, notes = _plan_task(pmaf, tmp_path, "dyn1_freq2", sc, [sc["random_vecs"]], ["--max-ticks", "400"])
    rows, traj = _oracle_node_run(oracle, scenes, sc, [sc["goal"]], [sc["random_vecs"]], 400, mult=2)
    _compare(data, notes, rows, traj)
    # (that the multiple really is in the rollouts: test_prediction_freq_multiple compares the predicted paths)
    assert len(data) == 400


@pytest.mark.parametrize("n_pops", [1, 6])
def test_move_real_with_zero_steps_does_not_mark_its_list_resident(pmaf, oracle, scenes, n_pops):
    """`pmaf_move_real(list, dt, steps = 0, ...)` staged the list -- which marks it resident -- although no
    manager launch ever read it; the next call with the SAME list then handed nothing over and planned on the old obstacles.
    Zero steps leave no trace in the reference either (the list is an argument of cfPlanner inside the steps loop,
    B/src/cf_manager.cpp:257-263). Six populations: the path where agent indices travel through a device buffer."""
    scs = [scenes.config_scene("C2", scene_id=i, dynamic=True) for i in range(n_pops)]
    sc = scs[0]
    starts = np.stack([s["start"] for s in scs])
    arg = scs if n_pops > 1 else sc
    hip = pmaf.PmafPlanner(arg, device=0, mgr_init_pos=starts if n_pops > 1 else sc["start"])
    oras = [oracle.OraclePlanner(s, mgr_init_pos=s["start"]) for s in scs]
    hip.set_initial_position(starts if n_pops > 1 else sc["start"])
    for o, s in zip(oras, scs):
        o.set_initial_position(s["start"])
    obs = np.stack([s["obstacles"] for s in scs])
    for t in range(3):
        bh = np.atleast_1d(hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]))
        bo = [o.tick(obs[p], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for p, o in enumerate(oras)]
        assert list(bh) == bo
    moved = np.stack([scenes.advance_live_obstacles(scenes.advance_live_obstacles(o)) for o in obs])
    hip.stop()
    hip.move_real(moved, sc["dt"], 0, bh)            # nothing happens, nothing is remembered
    for t in range(3):                               # ... so this tick hands `moved` over and plans on it
        bh = np.atleast_1d(hip.tick(moved, sc["dt"], sc["cost_gains"], sc["ws_limits"]))
        bo = [o.tick(moved[p], sc["dt"], sc["cost_gains"], sc["ws_limits"]) for p, o in enumerate(oras)]
        assert list(bh) == bo
        for p, o in enumerate(oras):
            _same(np.asarray(hip.real_state()[0]).reshape(n_pops, 3)[p], o.real_state()[0])
    hip.stop()
    ph, nh = hip.paths()
    for p, o in enumerate(oras):
        po, no = o.paths()
        _same(np.asarray(nh).reshape(n_pops, -1)[p], no)
        _same(np.asarray(ph).reshape((n_pops,) + np.asarray(po).shape)[p], po)
    hip.close()

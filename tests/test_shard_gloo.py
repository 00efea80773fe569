"""world_size-2 (and one world_size-4) gloo tests of the multi-GPU path (population sharding + winner
record all-gather + agent-range merge rule). Run on CPU: the rank-local
planner is the oracle (test-only compute stand-in for the kernels); everything
else is the product path -- shard.py over libpmaf_hip.so's communicator entry
points (pmaf_comm_init_host / pmaf_comm_allgather / pmaf_select_best) with
gloo as the host transport. The same code with an RCCL communicator and the
HIP planner runs in tests/test_shard_gpu.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_scenes, ticks, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    from oracle import orc
    pkg = graft.load_package()
    shard = __import__("pmaf_amd.shard", fromlist=["shard"])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = shard.make_comm(dist, world, rank, backend="host")
    mine = shard.partition_populations(n_scenes, world, rank)
    scs = [pkg.scenes.synthetic_scene(12, 60, 8, 8, s) for s in mine]
    planners = []
    for sc in scs:
        o = orc.OraclePlanner(sc, mgr_init_pos=sc["start"])
        o.set_initial_position(sc["start"])
        planners.append(o)
    cap = scs[0]["max_prediction_steps"]
    gathered = None
    for t in range(ticks):
        recs = []
        for sc, o in zip(scs, planners):
            o.stop()
            paths, n = o.paths()           # the rollouts this tick's selection scores
            b = o.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
            pos = o.real_state()[0]
            recs.append(shard.pack_winner_record(o.costs()[b], b, n[b], o.best_type(), paths[b], cap,
                                                 next_pos=pos, goal_dist=o.dist_from_goal()))
        gathered = shard.all_gather_winner_records(comm, np.stack(recs))
    q.put((rank, gathered.copy()))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n_scenes", [(2, 4), (4, 8), (8, 8)])
def test_population_sharding_and_winner_all_gather(world, n_scenes):
    import torch.multiprocessing as mp
    ticks = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scenes, ticks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every rank ends with the same gathered table
    for r in range(1, world):
        np.testing.assert_array_equal(results[0], results[r])
    # single-process reference of the same populations
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    from oracle import orc
    pkg = graft.load_package()
    shard = pkg.shard if hasattr(pkg, "shard") else __import__("pmaf_amd.shard", fromlist=["shard"])
    cap = 61
    for s in range(n_scenes):
        sc = pkg.scenes.synthetic_scene(12, 60, 8, 8, s)
        o = orc.OraclePlanner(sc, mgr_init_pos=sc["start"])
        o.set_initial_position(sc["start"])
        for t in range(ticks):
            paths, n = o.paths()
            b = o.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        rec = shard.unpack_winner_records(results[0][s % world, s // world], cap)[0]
        assert rec["index"] == b and rec["n_points"] == n[b] and rec["type"] == o.best_type()
        np.testing.assert_array_equal(rec["path"], paths[b, :n[b]])
        assert rec["cost"] == o.costs()[b]
        np.testing.assert_array_equal(rec["next_pos"], o.real_state()[0])
        assert rec["goal_dist"] == o.dist_from_goal()


def test_agent_range_merge_equals_single_population_selection(oracle, scenes):
    """splitting one population's cost vector over ranks and merging must give
    evaluateAgents' answer, including first-min ties and hysteresis"""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    shard = __import__("pmaf_amd.shard", fromlist=["shard"])
    sc = scenes.synthetic_scene(24, 80, 10, 8, 11)
    o = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
    o.set_initial_position(sc["start"])
    prev = None
    for t in range(40):
        b = o.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        costs = o.costs()
        for cuts in ([0, 24], [0, 8, 24], [0, 5, 11, 17, 24]):
            parts = [costs[a:b_] for a, b_ in zip(cuts[:-1], cuts[1:])]
            assert shard.merge_agent_ranges(parts, prev) == b
        prev = b
    # exact ties resolve to the lowest global index
    assert shard.merge_agent_ranges([np.array([3.0, 2.0]), np.array([2.0, 5.0])], None) == 1
    assert shard.merge_agent_ranges([np.array([3.0, 2.0]), np.array([1.9, 5.0])], 1) == 1  # 1.9 !< 0.9*2.0
    assert shard.merge_agent_ranges([np.array([3.0, 2.0]), np.array([1.7, 5.0])], 1) == 2


def _arm_scene(pkg, arm):
    s = pkg.scenes.synthetic_scene(16, 60, 10, 4, arm)
    s["start"] = np.array([-0.45, -0.12 if arm == 0 else 0.12, 0.7])
    s["goal"] = np.array([0.45, 0.10 if arm == 0 else -0.10, 0.7])
    return s


def _dual_worker(rank, world, port, ticks, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    from oracle import orc
    pkg = graft.load_package()
    shard = __import__("pmaf_amd.shard", fromlist=["shard"])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = shard.make_comm(dist, world, rank, backend="host")
    scs = [_arm_scene(pkg, a) for a in range(2)]
    sc = scs[rank]
    o = orc.OraclePlanner(sc, mgr_init_pos=sc["start"])
    o.set_initial_position(sc["start"])
    coupling = shard.DualArmCoupling(np.stack([s["obstacles"] for s in scs]), 0.1)
    pos = np.stack([s["start"] for s in scs])
    out = []
    for t in range(ticks):
        obs = coupling.coupled_obstacles(pos)
        o.tick(obs[rank], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        pos = shard.all_gather_positions(o.real_state()[0][None, :], comm)  # one 3-double exchange per tick
        out.append(pos.copy())
    q.put((rank, np.stack(out)))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dual_arm_one_population_per_rank_world2():
    """BASELINE config 4 layout: arm r on rank r, set-points exchanged with one
    all-gather per tick; must equal the single-process coupled run"""
    import torch.multiprocessing as mp
    world, ticks = 2, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dual_worker, args=(r, world, port, ticks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(results[0], results[1])
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    from oracle import orc
    pkg = graft.load_package()
    shard = __import__("pmaf_amd.shard", fromlist=["shard"])
    scs = [_arm_scene(pkg, a) for a in range(2)]
    oras = []
    for s in scs:
        o = orc.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    coupling = shard.DualArmCoupling(np.stack([s["obstacles"] for s in scs]), 0.1)
    pos = np.stack([s["start"] for s in scs])
    for t in range(ticks):
        obs = coupling.coupled_obstacles(pos)
        for i, o in enumerate(oras):
            o.tick(obs[i], scs[i]["dt"], scs[i]["cost_gains"], scs[i]["ws_limits"])
        pos = np.stack([o.real_state()[0] for o in oras])
        np.testing.assert_array_equal(results[0][t], pos)


def _range_worker(rank, world, port, ticks, q):
    """one agent-range shard per rank, a real cross-process gather"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    from oracle import orc
    pkg = graft.load_package()
    shard = __import__("pmaf_amd.shard", fromlist=["shard"])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = shard.make_comm(dist, world, rank, backend="host")
    sc = _range_scene(pkg)
    cuts = [0, 9, 24]   # ragged on purpose
    sh = shard.AgentRangeShard(orc.OraclePlanner, sc, cuts[rank], cuts[rank + 1], mgr_init_pos=sc["start"])
    sh.planner.set_initial_position(sc["start"])
    gather = shard.comm_gather(comm, max(b - a for a, b in zip(cuts[:-1], cuts[1:])), sc["obstacles"].shape[0])
    prev, out = None, []
    for t in range(ticks):
        best, pos = shard.sharded_tick([sh], prev, sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"], gather)
        out.append((best, pos.copy()))
        prev = best
    sh.planner.stop()
    q.put((rank, out, sh.planner.paths()))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def _range_scene(pkg):
    import json
    rec = dict(json.load(open(os.path.join(ROOT, "tests", "golden", "task_scenes.json")))["dual_arms_static1"])
    rec["n_agents"] = 24
    return pkg.scenes.scene_from_record(rec, "static1_24", horizon=400)


@pytest.mark.timeout(300)
def test_agent_range_shards_one_per_rank_world2():
    """ONE population split by agent range over two processes (SURVEY 8e
    fallback): per tick one all-gather of the cost vectors + one of the winner's
    heuristic, global selection by pmaf_select_best on every rank; set-points,
    best indices and paths must equal the unsharded oracle's"""
    import torch.multiprocessing as mp
    world, ticks = 2, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_range_worker, args=(r, world, port, ticks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = {r: (o, pa) for r, o, pa in got}
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    from oracle import orc
    pkg = graft.load_package()
    sc = _range_scene(pkg)
    ora = orc.OraclePlanner(sc, mgr_init_pos=sc["start"])
    ora.set_initial_position(sc["start"])
    seen = set()
    for t in range(ticks):
        bo = ora.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        for r in range(world):
            assert res[r][0][t][0] == bo
            np.testing.assert_array_equal(res[r][0][t][1], ora.real_state()[0])
        seen.add(bo)
    assert len(seen) >= 3
    po, no = ora.paths()
    cuts = [0, 9, 24]
    for r in range(world):
        ph, nh = res[r][1]
        np.testing.assert_array_equal(nh, no[cuts[r]:cuts[r + 1]])
        np.testing.assert_array_equal(ph, po[cuts[r]:cuts[r + 1]])

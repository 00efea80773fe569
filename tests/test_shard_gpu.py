"""GPU tests of the multi-GPU path through the C-ABI (include/pmaf.h "multi-GPU"):
an RCCL communicator created by libpmaf_hip.so itself (1 rank: the box has one
GPU), the stream-ordered winner-record all-gather, the per-tick exchange that
overlaps the rollout, and -- with two processes sharing GPU 0 over a
host-transport communicator (RCCL refuses two ranks on one device) -- the
population-sharded, dual-arm and agent-range layouts with the HIP planner on
every rank."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _portable_exp_oracle(oracle):
    oracle.set_exp_mode(1)
    yield
    oracle.set_exp_mode(0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scenes3(scenes):
    return [scenes.synthetic_scene(24, 90, 12, 5, s) for s in range(3)]


def test_rccl_one_rank_allgather_winners_stream_ordered(pmaf, scenes):
    """tick -> evaluate -> pmaf_allgather_winners (k_winner + ncclAllGather on the
    handle's stream, no host sync in between) with a 1-rank RCCL communicator
    created by the library: the received records equal the host getters bit for
    bit -- RCCL and libpmaf_hip.so share one HIP runtime in this process"""
    torch = pytest.importorskip("torch")
    scs = _scenes3(scenes)
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    comm = pmaf.PmafComm.rccl(1, 0, pmaf.PmafComm.unique_id(), 0)
    assert comm.world == 1 and comm.rank == 0
    sc = scs[0]
    obs = np.stack([s["obstacles"] for s in scs])
    for t in range(5):
        hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.stop()
    best = hip.evaluate(sc["cost_gains"], sc["ws_limits"])
    rec = hip.winner_record_doubles()
    assert rec == 8 + 3 * sc["max_prediction_steps"]
    buf = torch.full((1, 3, rec), -1.0, dtype=torch.float64, device="cuda:0")
    hip.allgather_winners(comm, buf.data_ptr(), buf.numel() * 8)
    hip.stop()   # = a stream-ordered consumer's view
    out = pmaf.shard.unpack_winner_records(buf.cpu().numpy(), sc["max_prediction_steps"])
    paths, n = hip.paths()
    costs = hip.costs()
    pos = hip.real_state()[0]
    dg = hip.dist_from_goal()
    for p in range(3):
        assert out[p]["index"] == best[p] and out[p]["n_points"] == n[p, best[p]]
        assert out[p]["cost"] == costs[p, best[p]]
        assert out[p]["type"] == hip.best()[0][p]
        np.testing.assert_array_equal(out[p]["path"], paths[p, best[p], :n[p, best[p]]])
        np.testing.assert_array_equal(out[p]["next_pos"], pos[p])
        assert out[p]["goal_dist"] == dg[p]
    # the control-plane all-gather of host data through the same communicator
    a = np.arange(7.0)
    np.testing.assert_array_equal(comm.allgather(a), a[None])
    hip.close()
    comm.close()


@pytest.mark.parametrize("transport", ["rccl", "host"])
def test_attached_exchange_every_tick_matches_getters_and_oracle(pmaf, oracle, scenes, transport):
    """pmaf_attach_comm: every pmaf_tick publishes its selection's winner records
    and all-gathers them on the exchange stream while the next rollout runs (the
    handle double-buffers its paths). Per tick the received table must hold the
    selected agent's path of the rollout that was scored, its cost, the new
    set-point; and the planner must stay bit-identical to the oracle."""
    scs = _scenes3(scenes)
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    oras = []
    for s in scs:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    if transport == "rccl":
        comm = pmaf.PmafComm.rccl(1, 0, pmaf.PmafComm.unique_id(), 0)
    else:
        comm = pmaf.PmafComm.host(1, 0, lambda b: b)
    hip.attach_comm(comm)
    hip.enable_winner_path()          # (ABI 5) the same path through pinned memory, with the path buffers alternating
    sc = scs[0]
    cap = sc["max_prediction_steps"]
    obs = np.stack([s["obstacles"] for s in scs])
    for t in range(30):
        hip.stop()
        prev_paths, prev_n = hip.paths()                   # what this tick's selection scores
        best = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        wpaths, wn, wa = hip.winner_path()
        for p in range(3):
            assert wa[p] == best[p] and wn[p] == prev_n[p, best[p]]
            np.testing.assert_array_equal(wpaths[p], prev_paths[p, best[p], :wn[p]])
        tab = hip.winners_wait()
        assert tab.shape == (1, 3, 8 + 3 * cap)
        recs = pmaf.shard.unpack_winner_records(tab[0], cap)
        costs = hip.costs()
        pos = hip.real_state()[0]
        for p in range(3):
            bo = oras[p].tick(obs[p], sc["dt"], sc["cost_gains"], sc["ws_limits"])
            assert best[p] == bo and recs[p]["index"] == bo
            assert recs[p]["n_points"] == prev_n[p, bo]
            np.testing.assert_array_equal(recs[p]["path"], prev_paths[p, bo, :prev_n[p, bo]])
            assert recs[p]["cost"] == costs[p, bo] == oras[p].costs()[bo]
            np.testing.assert_array_equal(recs[p]["next_pos"], pos[p])
            np.testing.assert_array_equal(pos[p], oras[p].real_state()[0])
    hip.stop()
    ph, nh = hip.paths()
    for p in range(3):
        po, no = oras[p].paths()
        np.testing.assert_array_equal(nh[p], no)
        np.testing.assert_array_equal(ph[p], po)
    times = hip.exchange_times_us()
    assert times.size == 30 and np.all(times >= 0)
    # checkpoint with an exchange attached, restore into a plain handle: identical continuation
    blob = hip.save_state()
    hip2 = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip2.load_state(blob)
    for t in range(5):
        b1 = hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        b2 = hip2.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        np.testing.assert_array_equal(b1, b2)
        np.testing.assert_array_equal(hip.real_state()[0], hip2.real_state()[0])
    # detach: back to one path buffer, still the same planner
    hip.attach_comm(None)
    for t in range(3):
        np.testing.assert_array_equal(hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]),
                                      hip2.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"]))
    hip.stop(); hip2.stop()
    np.testing.assert_array_equal(hip.paths()[0], hip2.paths()[0])
    with pytest.raises(pmaf.PmafError):
        hip.winners_wait()
    hip.close(); hip2.close(); comm.close()


def test_step_api_exchange_after_evaluate(pmaf, scenes):
    """stop / evaluate / move / reset / start with a communicator attached: the
    evaluate publishes the records; reset and restart must not disturb them"""
    scs = _scenes3(scenes)
    starts = np.stack([s["start"] for s in scs])
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    comm = pmaf.PmafComm.rccl(1, 0, pmaf.PmafComm.unique_id(), 0)
    hip.attach_comm(comm)
    sc = scs[0]
    cap = sc["max_prediction_steps"]
    obs = np.stack([s["obstacles"] for s in scs])
    hip.start()
    for t in range(6):
        hip.stop()
        paths, n = hip.paths()
        best = hip.evaluate(sc["cost_gains"], sc["ws_limits"])
        hip.move_real(obs, sc["dt"], 1, best)
        pos, vel, _ = hip.real_state()
        hip.reset_agents(pos, vel, obs)
        hip.start()
        recs = pmaf.shard.unpack_winner_records(hip.winners_wait()[0], cap)
        for p in range(3):
            assert recs[p]["index"] == best[p]
            np.testing.assert_array_equal(recs[p]["path"], paths[p, best[p], :n[p, best[p]]])
    hip.close(); comm.close()


def test_adopting_the_applications_own_nccl_communicator(pmaf, scenes):
    """pmaf_comm_from_rccl: the host application created its ncclComm_t itself (here: straight through librccl's C
    API, one rank); the library adopts it, exchanges through it and does not destroy it"""
    import ctypes as C
    pytest.importorskip("torch")   # (its bundled librccl is the one already mapped into the process)
    rccl = C.CDLL("librccl.so.1")
    uid = (C.c_ubyte * 128)()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0

    class NcclUniqueId(C.Structure):
        _fields_ = [("internal", C.c_ubyte * 128)]
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    u = NcclUniqueId()
    C.memmove(C.byref(u), uid, 128)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, u, 0) == 0
    pc = pmaf.PmafComm.from_rccl(comm.value, 0)
    assert pc.world == 1 and pc.rank == 0
    sc = scenes.synthetic_scene(16, 60, 8, 8, 1)
    hip = pmaf.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    hip.attach_comm(pc)
    for t in range(5):
        hip.stop()
        paths, n = hip.paths()
        b = hip.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        rec = pmaf.shard.unpack_winner_records(hip.winners_wait()[0], sc["max_prediction_steps"])[0]
        assert rec["index"] == b
        np.testing.assert_array_equal(rec["path"], paths[b, :n[b]])
    hip.attach_comm(None)
    hip.close()
    pc.close()
    # still usable by its owner after the library let go of it
    cnt = C.c_int(0)
    assert rccl.ncclCommCount(comm, C.byref(cnt)) == 0 and cnt.value == 1
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclCommDestroy(comm) == 0


# ---------------------------------------------------------------------------
# two processes on GPU 0 (host-transport communicator over gloo)
# ---------------------------------------------------------------------------
def _setup_worker(rank, world, port, multi_gpu=False, comm_ranks=None):
    """multi_gpu=False: every rank on GPU 0, host-transport communicator over gloo (RCCL refuses two ranks on one
    device); multi_gpu=True: rank r on GPU r, the library's own RCCL communicator (unique id through gloo).
    comm_ranks: the ranks the communicator spans (default all)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch  # noqa: F401  (before libpmaf_hip.so: one HIP runtime per process)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_ranks = list(range(world)) if comm_ranks is None else comm_ranks
    comm = None
    if not multi_gpu:
        assert comm_ranks == list(range(world))
        comm = pkg.shard.make_comm(dist, world, rank, backend="host")
    else:
        torch.cuda.set_device(rank)
        box = [pkg.PmafComm.unique_id() if rank == comm_ranks[0] else None]
        dist.broadcast_object_list(box, src=comm_ranks[0])
        if rank in comm_ranks:
            comm = pkg.PmafComm.rccl(len(comm_ranks), comm_ranks.index(rank), box[0], rank)
    return pkg, dist, comm


def _pop_worker(rank, world, port, n_scenes, ticks, q, multi_gpu=False):
    pkg, dist, comm = _setup_worker(rank, world, port, multi_gpu)
    mine = pkg.shard.partition_populations(n_scenes, world, rank)
    scs = [pkg.scenes.synthetic_scene(12, 60, 8, 8, s) for s in mine]
    starts = np.stack([s["start"] for s in scs])
    hip = pkg.PmafPlanner(scs, device=rank if multi_gpu else 0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    hip.attach_comm(comm)
    sc = scs[0]
    obs = np.stack([s["obstacles"] for s in scs])
    tab = None
    for t in range(ticks):
        hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        tab = hip.winners_wait()
    q.put((rank, tab.copy()))
    dist.barrier()
    hip.close()
    comm.close()
    dist.destroy_process_group()


def _spawn(target, world, *args, **kw):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=280) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


def _check_population_table(pmaf, oracle, scenes, res, world, n_scenes, ticks):
    for r in range(1, world):
        np.testing.assert_array_equal(res[0], res[r])
    cap = 61
    for s in range(n_scenes):
        sc = scenes.synthetic_scene(12, 60, 8, 8, s)
        o = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
        o.set_initial_position(sc["start"])
        for t in range(ticks):
            paths, n = o.paths()
            b = o.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        rec = pmaf.shard.unpack_winner_records(res[0][s % world, s // world], cap)[0]
        assert rec["index"] == b and rec["n_points"] == n[b] and rec["type"] == o.best_type()
        np.testing.assert_array_equal(rec["path"], paths[b, :n[b]])
        assert rec["cost"] == o.costs()[b]
        np.testing.assert_array_equal(rec["next_pos"], o.real_state()[0])


@pytest.mark.timeout(400)
def test_population_sharding_two_ranks_hip_planner(pmaf, oracle, scenes):
    """4 scenes over 2 ranks, HIP planner + attached exchange on every rank:
    both ranks end with the same table and it equals the unsharded oracles'"""
    world, n_scenes, ticks = 2, 4, 6
    res = dict(_spawn(_pop_worker, world, n_scenes, ticks))
    _check_population_table(pmaf, oracle, scenes, res, world, n_scenes, ticks)


def _n_gpus(pmaf):
    return min(pmaf.device_count(), 8)


@pytest.mark.timeout(900)
def test_rccl_population_sharding_n_ranks(pmaf, oracle, scenes):
    """One rank per GPU on min(hipGetDeviceCount(), 8) GPUs, the library's own
    RCCL communicator (ncclAllGather over xGMI) carrying the per-tick winner
    records: 2 scenes per rank, every rank's gathered table equals every other
    rank's and the per-population oracles. Needs >= 2 GPUs."""
    world = _n_gpus(pmaf)
    if world < 2:
        pytest.skip("needs >= 2 GPUs for an N-rank RCCL communicator (this box has %d); the same worker runs with the "
                    "host transport in test_population_sharding_two_ranks_hip_planner" % pmaf.device_count())
    n_scenes, ticks = 2 * world, 6
    res = dict(_spawn(_pop_worker, world, n_scenes, ticks, multi_gpu=True))
    _check_population_table(pmaf, oracle, scenes, res, world, n_scenes, ticks)


def _dual_peer_worker(rank, world, port, ticks, q, multi_gpu=False):
    """BASELINE C4, one arm per rank on ranks 0 / 1: set-points through the peer mailboxes (hipIpc), the path table
    through the 2-rank communicator; further ranks of the job only take part in the rendezvous"""
    pkg, dist, comm = _setup_worker(rank, world, port, multi_gpu, comm_ranks=[0, 1] if multi_gpu else None)
    arms = pkg.scenes.dual_arm_scenes(64, 150, 24)
    out, tab = None, None
    hip = None
    if rank < 2:
        sc = arms[rank]
        hip = pkg.PmafPlanner(sc, device=rank if multi_gpu else 0, mgr_init_pos=sc["start"])
        hip.set_initial_position(sc["start"])
        hip.attach_comm(comm)
    box = [None] * world
    dist.all_gather_object(box, hip.peer_export(2) if hip is not None else None)
    if hip is not None:
        hip.peer_connect(2, rank, box[:2])
        pkg.shard.couple_dual_arm_on_device(hip, 2, rank, np.stack([a["start"] for a in arms]))
    dist.barrier()
    if hip is not None:
        out = []
        for t in range(ticks):
            hip.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
            out.append(hip.real_state()[0].copy())
        hip.stop()
        tab = hip.winners_wait().copy()
        out = np.stack(out)
    q.put((rank, (out, tab)))
    dist.barrier()
    if hip is not None:
        hip.peer_disconnect()
        hip.attach_comm(None)
        hip.close()
    if comm is not None:
        comm.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_rccl_c4_one_arm_per_gpu(pmaf, oracle, scenes):
    """BASELINE config 4 on real GPUs: arm r on GPU r (ranks 0 / 1 of a
    min(hipGetDeviceCount(), 8)-rank job), set-points through hipIpc-mapped peer
    inboxes over xGMI, winner records through a 2-rank RCCL communicator; equals
    two coupled oracles bit for bit. Needs >= 2 GPUs (the one-GPU form of the same
    worker: tests/test_peer_gpu.py)."""
    world = _n_gpus(pmaf)
    if world < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d); two processes on one GPU run the same protocol in "
                    "tests/test_peer_gpu.py::test_peer_mailbox_two_processes_ipc_one_arm_per_rank" % pmaf.device_count())
    _check_dual_peer(pmaf, oracle, scenes, dict(_spawn(_dual_peer_worker, world, 200, multi_gpu=True)), world, 200)


@pytest.mark.timeout(400)
def test_c4_one_arm_per_rank_peer_mailboxes_two_ranks_one_gpu(pmaf, oracle, scenes):
    """the worker of test_rccl_c4_one_arm_per_gpu with both ranks on GPU 0 (host-transport communicator for the path
    table, hipIpc between the two processes for the set-points)"""
    _check_dual_peer(pmaf, oracle, scenes, dict(_spawn(_dual_peer_worker, 2, 200)), 2, 200)


def _check_dual_peer(pmaf, oracle, scenes, res, world, ticks):
    arms = scenes.dual_arm_scenes(64, 150, 24)
    oras = []
    for s in arms:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    coupling = pmaf.shard.DualArmCoupling(np.stack([s["obstacles"] for s in arms]), 0.1)
    pos = np.stack([s["start"] for s in arms])
    for t in range(ticks):
        obs = coupling.coupled_obstacles(pos)
        for i, o in enumerate(oras):
            o.tick(obs[i], arms[i]["dt"], arms[i]["cost_gains"], arms[i]["ws_limits"])
        pos = np.stack([o.real_state()[0] for o in oras])
        for r in (0, 1):
            np.testing.assert_array_equal(res[r][0][t], pos[r])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][1][:, 0, 4:7], pos)
    for r in range(2, world):
        assert res[r] == (None, None)


def _dual_worker(rank, world, port, ticks, q):
    pkg, dist, comm = _setup_worker(rank, world, port)
    arms = pkg.scenes.dual_arm_scenes(64, 150, 24)
    sc = arms[rank]
    hip = pkg.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    hip.attach_comm(comm)
    coupling = pkg.shard.DualArmCoupling(np.stack([s["obstacles"] for s in arms]), 0.1)
    pos = np.stack([s["start"] for s in arms])
    out = []
    for t in range(ticks):
        o = coupling.coupled_obstacles(pos)
        hip.tick(o[rank], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        pos = hip.winners_wait()[:, 0, 4:7].copy()     # both arms' set-points out of the winner records
        out.append(pos.copy())
    q.put((rank, np.stack(out)))
    dist.barrier()
    hip.close()
    comm.close()
    dist.destroy_process_group()


@pytest.mark.timeout(400)
def test_c4_dual_arm_one_arm_per_rank_hip_planner(pmaf, oracle, scenes):
    """BASELINE config 4 layout: arm r on rank r, the set-points travel in the
    per-tick winner records; equals two coupled oracles in one process"""
    world, ticks = 2, 200
    res = dict(_spawn(_dual_worker, world, ticks))
    np.testing.assert_array_equal(res[0], res[1])
    arms = scenes.dual_arm_scenes(64, 150, 24)
    oras = []
    for s in arms:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    coupling = pmaf.shard.DualArmCoupling(np.stack([s["obstacles"] for s in arms]), 0.1)
    pos = np.stack([s["start"] for s in arms])
    min_gap = 1e9
    for t in range(ticks):
        obs = coupling.coupled_obstacles(pos)
        for i, o in enumerate(oras):
            o.tick(obs[i], arms[i]["dt"], arms[i]["cost_gains"], arms[i]["ws_limits"])
        pos = np.stack([o.real_state()[0] for o in oras])
        np.testing.assert_array_equal(res[0][t], pos)
        min_gap = min(min_gap, np.linalg.norm(pos[0] - pos[1]))
    assert min_gap < arms[0]["detect_shell_rad"] + 0.15   # the spheres came into range: the coupling was exercised


def _range_scene(pkg):
    import json
    rec = dict(json.load(open(os.path.join(ROOT, "tests", "golden", "task_scenes.json")))["dual_arms_static1"])
    rec["n_agents"] = 24
    return pkg.scenes.scene_from_record(rec, "static1_24", horizon=400)


def _range_worker(rank, world, port, ticks, q):
    pkg, dist, comm = _setup_worker(rank, world, port)
    sc = _range_scene(pkg)
    cuts = [0, 9, 24]
    sh = pkg.shard.AgentRangeShard(pkg.PmafPlanner, sc, cuts[rank], cuts[rank + 1], device=0, mgr_init_pos=sc["start"])
    sh.planner.set_initial_position(sc["start"])
    gather = pkg.shard.comm_gather(comm, 15, sc["obstacles"].shape[0])
    prev, out = None, []
    for t in range(ticks):
        best, pos = pkg.shard.sharded_tick([sh], prev, sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"], gather)
        out.append((best, np.asarray(pos).copy()))
        prev = best
    sh.planner.stop()
    q.put((rank, out, sh.planner.paths()))
    dist.barrier()
    sh.planner.close()
    comm.close()
    dist.destroy_process_group()


@pytest.mark.timeout(400)
def test_agent_range_shards_two_ranks_hip_planner(pmaf, oracle, scenes):
    """ONE population split by agent range over two processes with the HIP
    planner on each: set-points, best indices, paths equal the unsharded oracle"""
    world, ticks = 2, 40
    got = _spawn(_range_worker, world, ticks)
    res = {r: (o, pa) for r, o, pa in got}
    sc = _range_scene(pmaf)
    ora = oracle.OraclePlanner(sc, mgr_init_pos=sc["start"])
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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("args,n_ranks", [
    ([], 2),                                   # default multi-GPU workload: C2 per rank + winner all-gather
    (["--config", "C5", "--shard", "--total-populations", "4"], 2),
    (["--config", "C4"], 2),
    ([], 4),
    (["--config", "C5", "--shard", "--total-populations", "8"], 4),
])
def test_bench_multi_rank_modes_on_one_gpu(args, n_ranks):
    """bench.py's N > 1 workloads, run as 2 or 4 ranks sharing GPU 0 with the host
    transport (test hooks PMAF_BENCH_BACKEND=gloo / PMAF_BENCH_SINGLE_DEVICE=1):
    the JSON line carries the collective's timing and per-rank tick times"""
    import json
    import subprocess
    env = dict(os.environ, PMAF_BENCH_BACKEND="gloo", PMAF_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(n_ranks), "--steps", "12", "--warmup", "3", "--min-seconds", "0.05", "--flop-ticks", "0"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == n_ranks and out["value"] > 0
    assert out["allgather_us"]["n"] >= 12 and out["allgather_us"]["median"] is not None
    assert len(out["tick_latency_us"]["per_rank_median"]) == n_ranks
    assert out["h_eff"] == out["config"]["horizon"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("hook", [{"PMAF_BENCH_C4_HOST_COUPLED": "1"}, {"PMAF_BENCH_FAIL_PEER": "connect"}, {"PMAF_BENCH_FAIL_PEER": "probe"}])
def test_bench_c4_couples_through_the_host_when_inboxes_cannot_be_shared(hook):
    """C4 one arm per rank on a runtime that cannot export fine-grained inboxes (forced with PMAF_BENCH_C4_HOST_COUPLED=1),
    or whose peer mailboxes fail to connect / fail their three probe ticks on some rank (PMAF_BENCH_FAIL_PEER): NOT a
    skipped record and not a dead job -- every rank learns of it in one reduction, each tick then waits for the winner
    table and takes the other arm's set-point out of it, and the record says so"""
    import json
    import subprocess
    env = dict(os.environ, PMAF_BENCH_BACKEND="gloo", PMAF_BENCH_SINGLE_DEVICE="1", **hook)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "12", "--warmup", "3", "--min-seconds", "0.05", "--flop-ticks", "0", "--config", "C4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["value"] and out["value"] > 0 and "skipped" not in out
    assert "THROUGH THE HOST" in out["config"]["workload"] and out["header_exchange_us"] is None
    if "PMAF_BENCH_FAIL_PEER" in hook:
        assert "peer mailboxes" in r.stderr and ("could not be connected" in r.stderr or "probe ticks" in r.stderr)
    assert out["allgather_us"]["n"] >= 12 and out["h_eff"] == out["config"]["horizon"]


@pytest.mark.timeout(900)
def test_bench_self_spawns_its_ranks_and_reports_every_config():
    """`python bench.py --gpus 2` WITHOUT a launcher (two ranks sharing GPU 0 under the test hooks): the script starts
    its ranks itself, the line says n_gpus = 2 and carries the C1 / C3 / C5-sharded / C4 sub-records -- C4 with the
    set-points through the peer mailboxes (header_exchange_us, no winners_wait on the tick path)"""
    import json
    import subprocess
    env = dict(os.environ, PMAF_BENCH_BACKEND="gloo", PMAF_BENCH_SINGLE_DEVICE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3",
                        "--min-seconds", "0.05", "--sub-seconds", "0.02", "--flop-ticks", "0"],
                       capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["collective_world"] == 2
    cfgs = out["configs"]
    assert set(cfgs) == {"C1", "C3", "C5_sharded", "C5_one_gpu", "C4", "task_static1", "C2_contracted", "C3_contracted", "C5_sharded_contracted"}
    assert all(cfgs[k]["arithmetic_policy"] == ("contracted" if k.endswith("_contracted") else "strict") for k in cfgs)
    # parity flags: strict = bit-exact; the contracted policy's tolerance contract does not hold on C5
    assert all(cfgs[k]["parity_met"] for k in cfgs if not k.endswith("_contracted"))
    assert cfgs["C2_contracted"]["parity_met"] and cfgs["C3_contracted"]["parity_met"] and not cfgs["C5_sharded_contracted"]["parity_met"]
    ts = cfgs.pop("task_static1")   # the shipped operating point: early stops allowed, no exchange
    assert ts["agents"] == 10 and ts["horizon"] == 1499 and ts["obstacles"] == 9 and 100 < ts["h_eff"] <= 1499
    assert ts["regime"]["tick_budget_ms"] == 10.0 and 0 < ts["regime"]["share_of_the_control_period"] < 1
    assert ts["kernel"] == "k_rollout_w64<1, 2, true, true>"
    for name, c in cfgs.items():
        assert c["rollouts_per_s"] > 0 and c["blocks"] >= 5 and c["h_eff"] == c["horizon"], name
        assert c["kernel"].startswith("k_rollout")
        if name == "C5_one_gpu":   # all eight scenes on rank 0's GPU alone: no exchange, one GPU
            assert c["allgather_us"] is None and c["gpus_used"] == 1 and c["populations_per_gpu"] == 8 and c["populations_total"] == 8
        else:
            assert c["allgather_us"]["n"] >= 10
    assert cfgs["C4"]["coupling"] == "peer mailboxes"
    assert cfgs["C5_sharded"]["populations_total"] == 8 and cfgs["C5_sharded"]["populations_per_gpu"] == 4
    # BASELINE C5's strong-scaling record at the top level, beside the weak-scaling `value`: measured at this N, the same eight
    # scenes on rank 0's GPU alone in the same job, and the ratio of the two (two ranks SHARING one GPU here: below 1)
    sc5 = out["scaling_c5"]
    assert sc5["n_gpus"] == 2 and sc5["scaling"] == "strong" and "weak" in sc5["read_this_for_scaling"]
    assert sc5["measured"]["rollouts_per_s"] == cfgs["C5_sharded"]["rollouts_per_s"]
    assert len(sc5["measured"]["per_gpu_tick_us"]) == 2 and len(sc5["measured"]["per_gpu_kernel_us"]) == 2     # per-GPU figures
    assert sc5["one_gpu_same_job"]["rollouts_per_s"] == cfgs["C5_one_gpu"]["rollouts_per_s"]
    assert abs(sc5["measured"]["speedup_vs_one_gpu_same_job"] - cfgs["C5_sharded"]["rollouts_per_s"] / cfgs["C5_one_gpu"]["rollouts_per_s"]) < 1e-12
    assert abs(sc5["measured"]["efficiency_vs_one_gpu_same_job"] * 2 - sc5["measured"]["speedup_vs_one_gpu_same_job"]) < 1e-12
    assert cfgs["C2_contracted"]["kernel"] == "k_rollout_w64<1, 3, true, true>"
    hx = cfgs["C4"]["header_exchange_us"]
    assert hx["n"] >= 50 and hx["wait_median"] is not None and cfgs["C4"]["gpus_used"] == 2
    # a launcher whose world differs from --gpus is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=ROOT)
    assert bad.returncode != 0 and "refusing" in bad.stderr


@pytest.mark.timeout(600)
def test_bench_single_rank_rccl_exchange():
    """bench.py with PMAF_BENCH_FORCE_DIST=1: one rank, torch.distributed (nccl)
    plus the library's own RCCL communicator and the per-tick exchange"""
    import json
    import subprocess
    env = dict(os.environ, PMAF_BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                        "--min-seconds", "0.1", "--cpu-seconds", "0", "--flop-ticks", "0"],
                       capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["allgather_us"]["n"] >= 20 and "RCCL" in out["config"]["workload"]


@pytest.mark.timeout(600)
def test_bench_one_gpu_line_carries_the_emulated_c5_scaling_curve():
    """the driver's own command on one GPU: next to `value` (BASELINE C2) the line holds `scaling_c5` -- BASELINE C5's
    measured one-GPU record and the 1 / 2 / 4 / 8-GPU curve emulated on this GPU (each rank's share of the eight scenes in a
    handle of its own; tick at N GPUs = the slowest rank's): VERDICT r5 item 1, SURVEY 8(e)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "PMAF_BENCH_FORCE_DIST")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--min-seconds", "0.1",
                        "--sub-seconds", "0.05", "--cpu-seconds", "0", "--flop-ticks", "0"],
                       capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["config"]["agents"] == 64
    sc5 = out["scaling_c5"]
    assert sc5["scaling"] == "strong" and sc5["n_gpus"] == 1 and sc5["measured"]["populations_per_gpu"] == 8
    assert sc5["measured"]["rollouts_per_s"] == out["configs"]["C5_sharded"]["rollouts_per_s"]
    by_n = sc5["prediction"]["predicted_by_n"]
    assert sorted(by_n) == ["1", "2", "4", "8"] and out["configs"]["C5_sharded"]["predicted_by_n"] == by_n
    assert [by_n[n]["populations_per_gpu"] for n in ("1", "2", "4", "8")] == [8, 4, 2, 1]
    assert [by_n[n]["lanes_per_agent"] for n in ("1", "2", "4", "8")] == [16, 32, 64, 64]            # pick_lpa per per-GPU load
    assert [len(by_n[n]["per_rank_ms_per_tick"]) for n in ("1", "2", "4", "8")] == [1, 2, 4, 8]      # every rank's share was run
    ms = [by_n[n]["ms_per_tick"] for n in ("1", "2", "4", "8")]
    assert ms[0] > ms[1] > ms[2] > ms[3] > 0.15                                                       # ... and floors on one chain
    assert all(by_n[n]["h_eff_min"] == 200.0 for n in by_n)                                           # full-horizon rollouts only
    eff = [by_n[n]["efficiency_vs_1gpu"] for n in ("1", "2", "4", "8")]
    assert eff[0] == 1.0 and eff[0] > eff[1] > eff[2] > eff[3] > 0.2
    assert abs(by_n["8"]["speedup_vs_1gpu"] - ms[0] / ms[3]) < 1e-9 and 2.0 < by_n["8"]["speedup_vs_1gpu"] < 4.0
    # the emulated one-GPU point IS the measured sub-record's workload: the two agree
    assert abs(sc5["measured"]["vs_predicted_ms_per_tick"] - 1.0) < 0.08


@pytest.mark.timeout(600)
def test_bench_rccl_bootstrap_failure_falls_back_to_host_transport():
    """if the library's RCCL communicator cannot be built, every rank switches to the host transport over a gloo
    side group and the run still measures the exchange (and says so)"""
    import json
    import subprocess
    env = dict(os.environ, PMAF_BENCH_FORCE_DIST="1", PMAF_BENCH_FAIL_RCCL="1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                        "--min-seconds", "0.1", "--cpu-seconds", "0", "--flop-ticks", "0"],
                       capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["allgather_us"]["n"] >= 20 and out["allgather_us"]["transport"] == "host"
    assert "host transport" in out["config"]["workload"]


@pytest.mark.timeout(600)
def test_exchange_lifecycle_does_not_leak(pmaf, scenes):
    """handle + communicator + attached exchange (second path buffer, record staging, exchange stream, events),
    created and destroyed in a loop: device memory returns to where it was and every cycle gives the same table;
    closing the handle with an exchange still in flight, and closing the communicator first, are both safe"""
    torch = pytest.importorskip("torch")
    scs = _scenes3(scenes)
    starts = np.stack([s["start"] for s in scs])
    sc = scs[0]
    obs = np.stack([s["obstacles"] for s in scs])

    def one(transport, wait):
        hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
        hip.set_initial_position(starts)
        comm = (pmaf.PmafComm.rccl(1, 0, pmaf.PmafComm.unique_id(), 0) if transport == "rccl"
                else pmaf.PmafComm.host(1, 0, lambda b: b))
        hip.attach_comm(comm)
        for _ in range(4):
            hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        tab = hip.winners_wait().copy() if wait else None   # wait=False: destroy with the exchange in flight
        hip.paths()                                         # the pinned host mirror of the paths as well
        hip.close()
        comm.close()
        return tab

    ref = {t: one(t, True) for t in ("rccl", "host")}
    np.testing.assert_array_equal(ref["rccl"], ref["host"])
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for i in range(12):
        for t in ("rccl", "host"):
            tab = one(t, i % 2 == 0)
            if tab is not None:
                np.testing.assert_array_equal(tab, ref[t])
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < 16 << 20, (free0, free1)
    # communicator closed before the handle it is attached to: the handle must be detached first, and says so
    hip = pmaf.PmafPlanner(scs, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    comm = pmaf.PmafComm.host(1, 0, lambda b: b)
    hip.attach_comm(comm)
    hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.attach_comm(None)
    comm.close()
    hip.tick(obs, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.close()

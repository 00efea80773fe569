"""GPU tests of the peer mailboxes (include/pmaf.h "peer mailboxes", ABI 3): the
header-only exchange WITHOUT a collective. The manager kernel of every tick
stores its populations' record headers straight into every rank's inbox
(device memory mapped into the peer processes with hipIpcOpenMemHandle) and a
coupled population takes its trailing repulsive obstacle from the header the
source population published one tick earlier -- BASELINE config 4's dual-arm
coupling with no host, stream or collective on the tick's control path.
Every layout is checked bit for bit against two coupled CPU oracles
(shard.DualArmCoupling on the host side)."""
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


def _coupled_oracles(oracle, shard, arms, ticks):
    """the reference run: two oracles coupled on the host; returns positions [ticks][2][3] and the closest approach"""
    oras = []
    for s in arms:
        o = oracle.OraclePlanner(s, mgr_init_pos=s["start"])
        o.set_initial_position(s["start"])
        oras.append(o)
    coupling = shard.DualArmCoupling(np.stack([s["obstacles"] for s in arms]), 0.1)
    pos = np.stack([s["start"] for s in arms])
    out, gap = [], 1e9
    for t in range(ticks):
        obs = coupling.coupled_obstacles(pos)
        for i, o in enumerate(oras):
            o.tick(obs[i], arms[i]["dt"], arms[i]["cost_gains"], arms[i]["ws_limits"])
        pos = np.stack([o.real_state()[0] for o in oras])
        out.append(pos.copy())
        gap = min(gap, np.linalg.norm(pos[0] - pos[1]))
    return np.stack(out), gap, oras


@pytest.mark.parametrize("pass_obstacles", [False, True])
def test_peer_mailbox_couples_two_populations_of_one_handle(pmaf, oracle, scenes, pass_obstacles):
    """world = 1: both arms are populations of ONE handle; each population's
    trailing obstacle comes out of the handle's own inbox (no IPC involved) --
    with the live obstacle list resident on the device and handed over per tick"""
    ticks = 200
    arms = scenes.dual_arm_scenes(64, 150, 24)
    starts = np.stack([s["start"] for s in arms])
    hip = pmaf.PmafPlanner(arms, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    pmaf.shard.connect_peers(hip, None, 1, 0)
    pmaf.shard.couple_dual_arm_on_device(hip, 1, 0, starts)
    sc = arms[0]
    obs = np.stack([s["obstacles"] for s in arms])    # trailing rows as shipped (100 m away): the kernel replaces them
    ref, gap, oras = _coupled_oracles(oracle, pmaf.shard, arms, ticks)
    for t in range(ticks):
        hip.tick(obs if pass_obstacles else None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        np.testing.assert_array_equal(hip.real_state()[0], ref[t])
    assert gap < sc["detect_shell_rad"] + 0.15        # the spheres came into range: the coupling mattered
    hip.stop()
    ph, nh = hip.paths()
    for i, o in enumerate(oras):
        po, no = o.paths()
        np.testing.assert_array_equal(nh[i], no)
        np.testing.assert_array_equal(ph[i], po)
    hd, sq = hip.peer_read()
    assert (sq == ticks).all()
    np.testing.assert_array_equal(hd[0, :, 4:7], ref[-1])
    w, pb = hip.peer_times_us()
    assert w.size == ticks and (w >= 0).all() and (pb >= 0).all()
    print("peer mailbox, one handle: header wait median %.2f us, publish median %.2f us" % (np.median(w), np.median(pb)))
    hip.peer_disconnect()
    hip.close()


def test_peer_mailbox_two_handles_in_one_process(pmaf, oracle, scenes):
    """world = 2 with both "ranks" in this process (one host driving two
    handles): the peers' inboxes are used through their own pointers"""
    ticks = 120
    arms = scenes.dual_arm_scenes(48, 120, 20)
    hs = []
    for s in arms:
        h = pmaf.PmafPlanner(s, device=0, mgr_init_pos=s["start"])
        h.set_initial_position(s["start"])
        hs.append(h)
    handles = [h.peer_export(2) for h in hs]
    starts = np.stack([s["start"] for s in arms])
    for r, h in enumerate(hs):
        h.peer_connect(2, r, handles)
        pmaf.shard.couple_dual_arm_on_device(h, 2, r, starts)
    ref, gap, _ = _coupled_oracles(oracle, pmaf.shard, arms, ticks)
    sc = arms[0]
    for t in range(ticks):
        for r, h in enumerate(hs):
            h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        for r, h in enumerate(hs):
            np.testing.assert_array_equal(hs[r].real_state()[0], ref[t, r])
    for h in hs:
        h.stop()
    for h in hs:
        h.peer_disconnect()
        h.close()


def test_peer_mailbox_missing_header_times_out(pmaf, scenes, monkeypatch):
    """a coupled header that never arrives: the manager kernel gives up after
    PMAF_PEER_TIMEOUT_S and pmaf_tick reports it instead of hanging the GPU"""
    monkeypatch.setenv("PMAF_PEER_TIMEOUT_S", "0.2")
    arms = scenes.dual_arm_scenes(16, 40, 8)
    h = pmaf.PmafPlanner(arms[0], device=0, mgr_init_pos=arms[0]["start"])
    h.set_initial_position(arms[0]["start"])
    h2 = pmaf.PmafPlanner(arms[1], device=0, mgr_init_pos=arms[1]["start"])
    handles = [h.peer_export(2), h2.peer_export(2)]
    h.peer_connect(2, 0, handles)
    h.peer_couple(0, 1, 0, 0.1)            # no init_pos, and "rank 1" never ticks
    sc = arms[0]
    with pytest.raises(pmaf.PmafError) as e:
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    assert e.value.code == -2 and "did not arrive" in str(e.value)
    with pytest.raises(pmaf.PmafError):
        h.peer_couple(0, 1, 0, 0.1, arms[1]["start"])   # init_pos only before the first tick
    h.stop()
    h.peer_disconnect()
    h.close(); h2.close()


# ---------------------------------------------------------------------------
# two processes sharing GPU 0: the inboxes really travel as hipIpc handles
# ---------------------------------------------------------------------------
def _ipc_worker(rank, world, port, ticks, with_rccl_table, q):
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401  (before libpmaf_hip.so: one HIP runtime per process)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arms = pkg.scenes.dual_arm_scenes(64, 150, 24)
    sc = arms[rank]
    hip = pkg.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    hip.set_initial_position(sc["start"])
    comm = None
    if with_rccl_table:   # the path table keeps travelling through the (host-transport) all-gather, off the tick path
        comm = pkg.shard.make_comm(dist, world, rank, backend="host")
        hip.attach_comm(comm)
    pkg.shard.connect_peers(hip, dist, world, rank)
    pkg.shard.couple_dual_arm_on_device(hip, world, rank, np.stack([a["start"] for a in arms]))
    dist.barrier()
    out = []
    for t in range(ticks):
        hip.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])   # nothing but the tick on the control path
        out.append(hip.real_state()[0].copy())
    hip.stop()
    tab = hip.winners_wait().copy() if comm is not None else None
    w, pb = hip.peer_times_us()
    hd, sq = hip.peer_read()
    q.put((rank, np.stack(out), tab, float(np.median(w)), float(np.percentile(w, 99)), float(np.median(pb)), hd, sq))
    dist.barrier()          # nobody unmaps while a peer may still store into its inbox
    hip.peer_disconnect()
    if comm is not None:
        hip.attach_comm(None)
        comm.close()
    hip.close()
    dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize("with_table", [False, True])
def test_peer_mailbox_two_processes_ipc_one_arm_per_rank(pmaf, oracle, scenes, with_table):
    """BASELINE config 4, one arm per PROCESS (two ranks sharing GPU 0): the
    set-points travel through hipIpc-mapped inboxes only; bit-identical to two
    coupled oracles for 200 ticks, also with the winner-record all-gather
    attached beside it (then the table of the last tick holds both arms' records)"""
    import torch.multiprocessing as mp
    world, ticks = 2, 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ipc_worker, args=(r, world, port, ticks, with_table, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=300)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    arms = scenes.dual_arm_scenes(64, 150, 24)
    ref, gap, oras = _coupled_oracles(oracle, pmaf.shard, arms, ticks)
    for r in range(world):
        np.testing.assert_array_equal(got[r][0], ref[:, r])
        hd, sq = got[r][5], got[r][6]
        assert (sq == ticks).all()
        np.testing.assert_array_equal(hd[:, 0, 4:7], ref[-1])      # every rank's inbox holds both arms' last headers
    assert gap < arms[0]["detect_shell_rad"] + 0.15
    if with_table:
        np.testing.assert_array_equal(got[0][1], got[1][1])
        np.testing.assert_array_equal(got[0][1][:, 0, 4:7], ref[-1])
    print("peer mailbox over hipIpc: header wait median %.2f / %.2f us (p99 %.1f / %.1f), publish median %.2f / %.2f us"
          % (got[0][2], got[1][2], got[0][3], got[1][3], got[0][4], got[1][4]))


def test_one_way_coupling_is_rejected(pmaf, scenes):
    """the two parity slots per source rest on PAIRWISE MUTUAL couplings (a source cannot overwrite header
    t-1 before it has read the consumer's header t). A one-way coupling has no such back-pressure: the consumer sees, in
    the first kernel-written header of its source, that the source is not coupled back, and pmaf_tick fails with
    PMAF_ERR_STATE at once instead of timing out some tick later"""
    arms = scenes.dual_arm_scenes(16, 60, 8)
    starts = np.stack([s["start"] for s in arms])
    hip = pmaf.PmafPlanner(arms, device=0, mgr_init_pos=starts)
    hip.set_initial_position(starts)
    pmaf.shard.connect_peers(hip, None, 1, 0)
    info = hip.peer_info()
    assert info["world"] == 1 and isinstance(info["fine_grained"], bool)
    hip.peer_couple(0, 0, 1, 0.1, init_pos=starts[1])      # population 0 follows population 1 -- and not vice versa
    sc = arms[0]
    hip.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])          # tick 1 consumes the host-written header
    with pytest.raises(pmaf.PmafError) as ei:
        hip.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])      # tick 2: the source's own header says "uncoupled"
    assert ei.value.code == -3 and "pairwise mutual" in str(ei.value)
    # made mutual, the pair works (the coupling restarts from host-written headers after a reconnect)
    hip.peer_disconnect()
    pmaf.shard.connect_peers(hip, None, 1, 0)
    pmaf.shard.couple_dual_arm_on_device(hip, 1, 0, np.asarray(hip.real_state()[0]))
    for _ in range(5):
        hip.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
    hip.peer_disconnect()
    hip.close()

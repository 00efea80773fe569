"""Multi-GPU sharding of the planner tick -- Python orchestration
over the C-ABI's multi-GPU entry points (include/pmaf.h: pmaf_comm_*,
pmaf_attach_comm, pmaf_winners_wait, pmaf_select_best). Used by bench.py and
the tests; a C++ host calls the same entry points directly.

The path shards by POPULATION: agents of different populations (scenes, arms,
goal sweeps) never interact, so rank r plans the populations
{s : s % world == r} with its own handle and no collective on the rollout's
data path. Only when a caller needs every population's winning trajectory on
every rank (the dual-arm extension where each arm treats the other arm's
end effector as its repulsive obstacle, a goal sweep's global pick) the
fixed-size winner records are exchanged with ONE all-gather per tick --
ncclAllGather (RCCL over xGMI) enqueued by libpmaf_hip.so on a second stream
behind the selection kernel, overlapping the next rollout; a host-transport
communicator (gloo / MPI callback) takes its place in the CPU tests. Records
are (8 + 3*cap) doubles: cost, agent index, n_points, agent type, the real
agent's next position[3], its goal distance, path[cap][3] (<= 12 KB at cap
501), so the collective is latency-bound; nothing here is sized by link
bandwidth.

For the fallback of ONE population split by agent range, merge_agent_ranges()
applies CfManager::evaluateAgents' selection rule (B/src/cf_manager.cpp:336-353;
pmaf_select_best) to the gathered per-rank costs: lowest cost, ties to the
lowest global index, then the 0.9 hysteresis against the previous global best.
"""
import numpy as np

from . import planner as _planner

WINNER_HDR = 8  # PMAF_WINNER_RECORD_HEADER


def partition_populations(n_populations, world, rank):
    """population ids owned by `rank` (round robin, like SURVEY 8e 'scene s on GPU s mod G')"""
    return [s for s in range(n_populations) if s % world == rank]


def record_doubles(cap):
    return WINNER_HDR + 3 * cap


def pack_winner_record(cost, idx, n_points, agent_type, path, cap, next_pos=(0.0, 0.0, 0.0), goal_dist=0.0):
    """host-side packing of one winner record (same layout as k_manager / k_winner write)"""
    rec = np.zeros(record_doubles(cap))
    rec[0], rec[1], rec[2], rec[3] = cost, idx, n_points, agent_type
    rec[4:7] = next_pos
    rec[7] = goal_dist
    rec[WINNER_HDR:WINNER_HDR + 3 * n_points] = np.asarray(path)[:n_points].reshape(-1)
    return rec


def unpack_winner_records(flat, cap):
    """[dict(cost, index, n_points, type, next_pos, goal_dist, path[n_points][3]), ...] from gathered records"""
    flat = np.asarray(flat, dtype=np.float64).reshape(-1, record_doubles(cap))
    out = []
    for r in flat:
        n = int(r[2])
        out.append(dict(cost=float(r[0]), index=int(r[1]), n_points=n, type=int(r[3]),
                        next_pos=r[4:7].copy(), goal_dist=float(r[7]),
                        path=r[WINNER_HDR:WINNER_HDR + 3 * n].reshape(n, 3).copy()))
    return out


def torch_host_allgather(dist):
    """the callable of a host-transport communicator (PmafComm.host) over a
    torch.distributed CPU process group (gloo): bytes in, world * bytes out"""
    def allgather(send_u8):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(send_u8))
        out = torch.empty(dist.get_world_size() * t.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, t)
        return out.numpy()
    return allgather


def make_comm(dist, world, rank, backend="rccl", device=-1):
    """Communicator of a multi-process run bootstrapped over torch.distributed:
    "rccl" -- rank 0 draws the ncclUniqueId, the 128 bytes travel through the
    process group, every rank calls ncclCommInitRank (pmaf_comm_init_rccl);
    "host" -- all-gathers run on the process group itself (gloo; CPU tests and
    several ranks sharing one GPU)."""
    if backend == "host":
        return _planner.PmafComm.host(world, rank, torch_host_allgather(dist))
    box = [_planner.PmafComm.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return _planner.PmafComm.rccl(world, rank, box[0], device)


def all_gather_winner_records(comm, local_records):
    """local_records [P_local][rec] host array, same P_local on every rank ->
    [world][P_local][rec]; population s of the global numbering is
    out[s % world, s // world]. (Host-side exchange through the communicator;
    handles with an attached communicator exchange on the device instead:
    PmafPlanner.winners_wait.)"""
    return comm.allgather(np.ascontiguousarray(local_records, dtype=np.float64))


def merge_agent_ranges(costs_per_rank, prev_best_global):
    """Global selection for ONE population whose agents are split into
    contiguous ranges over ranks. costs_per_rank: list of 1-D arrays in rank
    order. prev_best_global: previous best global index or None. Returns the
    new best global index (evaluateAgents semantics, cf_manager.cpp:336-353,
    evaluated by the library's pmaf_select_best)."""
    costs = np.concatenate([np.asarray(c, dtype=np.float64) for c in costs_per_rank])
    return _planner.select_best(costs, prev_best_global)


class DualArmCoupling:
    """BASELINE config 4 (build-defined): two independent agent
    populations, one per arm; each arm's trailing repulsive obstacle (the
    reference's "self collision" sphere, README.md:80) follows the OTHER arm's
    current end-effector position. Works on any set of planner objects exposing
    tick()/real_state(): two populations in one handle on one GPU, or one
    population per rank with the set-points exchanged by a 3-double all-gather.

    tick(k) uses the other arm's position after tick k-1 (the positions both
    arms published last), so the arms stay independent within a tick."""

    def __init__(self, obstacles, self_collision_radius=0.1):
        # obstacles: [2][n_obs][7], row -1 of each arm is replaced every tick
        self.obstacles = np.array(obstacles, dtype=np.float64, copy=True)
        self.radius = self_collision_radius

    def coupled_obstacles(self, ee_positions, advance=None):
        """ee_positions [2][3] = both arms' current real-agent positions"""
        obs = self.obstacles
        if advance is not None:
            obs = np.stack([advance(o) for o in obs])
        for arm in (0, 1):
            obs[arm, -1, 0:3] = ee_positions[1 - arm]
            obs[arm, -1, 3:6] = 0.0
            obs[arm, -1, 6] = self.radius
        self.obstacles = obs
        return obs


def connect_peers(planner, dist, world, rank):
    """Peer mailboxes of a multi-process run (include/pmaf.h "peer mailboxes"):
    every rank exports its handle's inbox, the 128-byte handles travel through
    the process group, every rank maps every peer's inbox
    (hipIpcOpenMemHandle). dist = torch.distributed (None for world == 1)."""
    mine = planner.peer_export(world)
    box = [mine]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, mine)
    planner.peer_connect(world, rank, box)


def couple_dual_arm_on_device(planner, world, rank, starts, radius=0.1):
    """BASELINE config 4 with the set-points travelling through the peer
    mailboxes instead of the host: arm a's trailing obstacle follows arm 1 - a.
    world == 2: one arm per rank (population 0 of each); world == 1: both arms
    are populations 0 / 1 of this handle. starts [2][3] = the arms' start
    positions (what the first tick sees), as shard.DualArmCoupling."""
    starts = np.asarray(starts, dtype=np.float64).reshape(2, 3)
    if world == 2:
        planner.peer_couple(0, 1 - rank, 0, radius, starts[1 - rank])
    else:
        planner.peer_couple(0, rank, 1, radius, starts[1])
        planner.peer_couple(1, rank, 0, radius, starts[0])


def all_gather_positions(local_pos, comm):
    """[P_local][3] set-points of every rank -> [world*P_local][3] (rank-major), one small all-gather"""
    local = np.ascontiguousarray(local_pos, dtype=np.float64).reshape(-1, 3)
    if comm is None or comm.world == 1:
        return local
    return comm.allgather(local).reshape(-1, 3)


class AgentRangeShard:
    """One rank's share of ONE population split by contiguous agent range
    (the fallback of SURVEY.md 8e for a population too large for one GPU).

    Every rank owns a planner holding agents [a0, a1) of the population (their
    types, gains and Random vectors) and a replica of the real agent. Per tick:
      1. local costs of the finished rollouts (pmaf_evaluate on the shard);
      2. ONE all-gather of the per-rank cost vectors (+ the local candidates'
         type and Random vectors, needed by whoever wins);
      3. the global selection, identical on every rank: first minimum over the
         concatenated costs, 0.9 hysteresis against the previous global best
         (merge_agent_ranges = CfManager::evaluateAgents :336-353);
      4. the winner's heuristic is installed as the shard's best-agent copy
         (pmaf_set_best) and every rank steps its replica of the real agent
         (bit-identical inputs -> bit-identical replicas), resets and restarts
         its agents.
    Gains must be uniform over the agents (they are in the reference:
    B/src/panda_bimanual_control.cpp:464-468), because the real step takes the
    best agent's gains.

    `gather(list_of_arrays_per_local_shard) -> list over ALL shards in rank
    order` abstracts the collective: torch.distributed all_gather_object /
    all_gather in a multi-process run, identity when one process drives all
    shards (tests)."""

    def __init__(self, planner_cls, scene, a0, a1, **planner_kw):
        from . import scenes as _scenes
        n = int(scene["n_agents"])
        types = np.asarray(scene.get("agent_types", _scenes.default_agent_types(n)), dtype=np.int32)
        sub = dict(scene)
        sub["n_agents"] = a1 - a0
        sub["agent_types"] = types[a0:a1].copy()
        sub["random_vecs"] = np.ascontiguousarray(scene["random_vecs"][a0:a1])
        for k in ("k_attr", "k_circ", "k_repel", "k_damp"):
            g = np.broadcast_to(np.asarray(scene[k], dtype=np.float64), (n,))
            if not np.all(g == g[0]):
                raise ValueError("agent-range sharding needs uniform gains (%s)" % k)
            sub[k] = float(g[0])
        self.a0, self.a1 = a0, a1
        self.types = sub["agent_types"]
        self.rand = sub["random_vecs"]
        self.scene = sub
        self.planner = planner_cls(sub, **planner_kw)

    # -- step 1 / 2: what this shard contributes to the all-gather
    def local_costs(self, cost_gains, ws):
        self.planner.stop()
        self.planner.evaluate(cost_gains, ws)  # local selection result is discarded
        return np.asarray(self.planner.costs(), dtype=np.float64)

    def candidate(self, local_index):
        """(type, random vectors) of one of this shard's agents"""
        return int(self.types[local_index]), self.rand[local_index]

    # -- step 4
    def apply_global_best(self, best_type, best_rand, obstacles, dt):
        p = self.planner
        p.set_best(1, best_type, best_rand)   # id only has to be a valid local id: selection is global
        p.move_real(obstacles, dt, 1, 0)      # uniform gains: any local agent's
        pos, vel, _ = p.real_state()
        p.reset_agents(pos, vel, obstacles)
        p.start()
        return pos


def comm_gather(comm, max_agents_per_rank, n_obs):
    """`gather` of sharded_tick over a communicator, ONE shard per rank: cost
    vectors travel padded to max_agents_per_rank (padding = DBL_MAX never wins
    the strict-< argmin and is stripped again), candidate infos as
    (valid, type, random vectors[n_obs][3])."""
    big = np.finfo(np.float64).max

    def gather(local):
        item = local[0]
        if item is None or isinstance(item, tuple):  # candidate info of the winner's owner
            buf = np.zeros(2 + 3 * n_obs)
            if item is not None:
                buf[0], buf[1] = 1.0, float(item[0])
                buf[2:] = np.asarray(item[1], dtype=np.float64).reshape(-1)
            allb = comm.allgather(buf)
            return [(int(b[1]), b[2:].reshape(n_obs, 3).copy()) if b[0] == 1.0 else None for b in allb]
        c = np.asarray(item, dtype=np.float64)
        buf = np.full(1 + max_agents_per_rank, big)
        buf[0] = c.size
        buf[1:1 + c.size] = c
        allb = comm.allgather(buf)
        return [b[1:1 + int(b[0])].copy() for b in allb]
    return gather


def sharded_tick(shards, prev_best_global, obstacles, dt, cost_gains, ws, gather=None):
    """One planner tick of a population split over `shards` (all shards of this
    process; with one shard per rank pass gather = comm_gather(...), an
    all-gather across ranks). Returns (global best index, next real position)."""
    local = [s.local_costs(cost_gains, ws) for s in shards]
    costs = gather(local) if gather is not None else local
    best = merge_agent_ranges(costs, prev_best_global)
    # the owner of the winner publishes its heuristic (type + Random vectors)
    offs = np.cumsum([0] + [len(c) for c in costs])
    owner = int(np.searchsorted(offs, best, side="right") - 1)
    info_local = []
    for s in shards:
        info_local.append(s.candidate(best - s.a0) if s.a0 <= best < s.a1 else None)
    infos = gather(info_local) if gather is not None else info_local
    btype, brand = infos[owner]
    pos = None
    for s in shards:
        pos = s.apply_global_best(btype, brand, obstacles, dt)
    return best, pos

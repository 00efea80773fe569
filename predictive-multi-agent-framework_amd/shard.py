"""Multi-GPU sharding of the planner tick.

The path shards by POPULATION: agents of different populations (scenes, arms,
goal sweeps) never interact, so rank r plans the populations
{s : s % world == r} with its own PmafPlanner and no data-path collective.
Only when a caller needs every population's winning trajectory on every rank
(e.g. the dual-arm extension where each arm treats the other arm's predicted
path as its repulsive obstacle) the fixed-size winner records are exchanged
with ONE all-gather per tick -- RCCL over xGMI when the tensors are on the GPU
(torch.distributed backend "nccl"), gloo in the CPU tests. Records are
(4 + 3*cap) doubles: cost, agent index, n_points, agent type, path[cap][3]
(<= 12 KB at cap 501), so the collective is latency-bound; nothing here is
sized by link bandwidth.

For the fallback of ONE population split by agent range, merge_agent_ranges()
applies CfManager::evaluateAgents' selection rule (B/src/cf_manager.cpp:336-353)
to the gathered per-rank costs: lowest cost, ties to the lowest global index,
then the 0.9 hysteresis against the previous global best.
"""
import numpy as np


def partition_populations(n_populations, world, rank):
    """population ids owned by `rank` (round robin, like SURVEY 8e 'scene s on GPU s mod G')"""
    return [s for s in range(n_populations) if s % world == rank]


def record_doubles(cap):
    return 4 + 3 * cap


def pack_winner_record(cost, idx, n_points, agent_type, path, cap):
    """host-side packing of one winner record (same layout as k_winner)"""
    rec = np.zeros(record_doubles(cap))
    rec[0], rec[1], rec[2], rec[3] = cost, idx, n_points, agent_type
    rec[4:4 + 3 * n_points] = np.asarray(path)[:n_points].reshape(-1)
    return rec


def unpack_winner_records(flat, cap):
    """[(cost, idx, n_points, type, path[n_points][3]), ...] from gathered records"""
    flat = np.asarray(flat, dtype=np.float64).reshape(-1, record_doubles(cap))
    out = []
    for r in flat:
        n = int(r[2])
        out.append(dict(cost=float(r[0]), index=int(r[1]), n_points=n, type=int(r[3]),
                        path=r[4:4 + 3 * n].reshape(n, 3).copy()))
    return out


def all_gather_winner_records(local_records, dist, world):
    """local_records: torch tensor [P_local, rec] (cuda -> RCCL, cpu -> gloo),
    same P_local on every rank. Returns [world, P_local, rec]; population s of
    the global numbering is out[s % world, s // world]."""
    import torch
    local = local_records.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out.view((world,) + tuple(local.shape))


def gather_from_planner(planner, dist, world):
    """winner records of a PmafPlanner straight from device memory: k_winner
    writes into a CUDA tensor on the planner's stream, the all-gather runs after
    a stream sync. Returns a [world, P, rec] CUDA tensor."""
    import torch
    rec = planner.winner_record_doubles()
    buf = torch.empty((planner.P, rec), dtype=torch.float64, device="cuda")
    planner.write_winner_records(buf.data_ptr(), buf.numel() * 8)
    planner.stop()
    return all_gather_winner_records(buf, dist, world)


def merge_agent_ranges(costs_per_rank, prev_best_global):
    """Global selection for ONE population whose agents are split into
    contiguous ranges over ranks. costs_per_rank: list of 1-D arrays in rank
    order. prev_best_global: previous best global index or None. Returns the
    new best global index (evaluateAgents semantics, cf_manager.cpp:336-353)."""
    costs = np.concatenate([np.asarray(c, dtype=np.float64) for c in costs_per_rank])
    min_idx, min_cost = 0, np.finfo(np.float64).max
    for i, c in enumerate(costs):
        if c < min_cost:
            min_cost, min_idx = c, i
    if prev_best_global is not None:
        if costs[min_idx] < 0.9 * costs[prev_best_global]:
            return min_idx
        return prev_best_global
    return min_idx


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


def all_gather_positions(local_pos, dist, world):
    """[P_local][3] set-points of every rank -> [world*P_local][3] (rank-major)"""
    import torch
    local = torch.as_tensor(np.ascontiguousarray(local_pos, dtype=np.float64))
    out = torch.empty((world * local.shape[0], 3), dtype=torch.float64)
    if dist is None or world == 1:
        return local.numpy()
    dist.all_gather_into_tensor(out, local)
    return out.numpy()


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


def sharded_tick(shards, prev_best_global, obstacles, dt, cost_gains, ws, gather=None):
    """One planner tick of a population split over `shards` (all shards of this
    process; with one shard per rank pass gather = an all-gather of Python
    objects across ranks). Returns (global best index, next real position)."""
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

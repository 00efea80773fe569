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

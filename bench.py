#!/usr/bin/env python3
"""bench.py -- planner-tick throughput of the MI355X-native circular-field
planner (BASELINE.json metric: agent-rollouts/s + planner tick latency).

One "step" = one planner tick (pmaf_tick: stop -> evaluateAgents ->
moveRealEEAgent -> resetEEAgents -> startPrediction) of every population of
the workload. Obstacles / agent state are resident in HBM before the timed
region; ticks are issued back to back, each returning best index + next
set-point to the host (the real per-tick API, not a batched open-loop shortcut).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU. Launched by `torch.distributed.run` the ranks come
from the environment; WITHOUT a launcher (`python bench.py --gpus 8`) the
script starts its N ranks itself (re-executes under torch.distributed.run on
127.0.0.1). --gpus must equal the world size and must not exceed
hipGetDeviceCount() -- anything else is an error, never a silent 1-GPU run.

The line's `value` is the HEADLINE workload: BASELINE C2 (64 agents x 200 steps
x 32 obstacles), one population per GPU ("weak" scaling: every rank plans the
same scene), with -- for N > 1 -- the winner records all-gathered once per tick
by ncclAllGather (RCCL over xGMI), enqueued by libpmaf_hip.so beside the next
rollout. The same invocation then times BASELINE's other configurations and
reports them under "configs" (>= 5 blocks each, same timing rules):
  C1          static1 scene, 16 agents x 100 steps (per-rank replicas)
  C3          256 agents x 500 steps x 128 obstacles (per-rank replicas)
  C5_sharded  8 goal/obstacle scenes x 1024 agents partitioned over the N ranks,
              scene s on rank s % N: the strong-scaling curve of config 5 as the
              driver runs N = 1, 2, 4, 8
  C4          dual arm, 2 x 256 agents, each arm's repulsive sphere follows the
              other arm's set-point THROUGH THE PEER MAILBOXES (no collective and
              no host on the control path; include/pmaf.h): N = 1 both arms in
              one handle, N >= 2 one arm per GPU on ranks 0 / 1 (hipIpc-mapped
              inboxes over xGMI), the path table still all-gathered beside it
--only-headline skips them; --config / --shard / --populations / --dynamic
select another headline workload (then no sub-configurations are run).

Timing: W warm-up ticks, then blocks of K ticks, each block bracketed by a
barrier + device synchronisation on both sides; blocks are repeated until
--min-seconds have been timed (and at least --min-blocks blocks) and the
MEDIAN block (max over ranks per block) is reported.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6   # 256 CU x 2.4 GHz x 128 flop/clk (vector FP64)


def algorithmic_bytes_per_tick(N, H, n_obs):
    """path write N*(H+1)*24 B + per-agent results N*32 B;
    reads (M+1)*56 B obstacles + N*48 B agent state."""
    return N * (H + 1) * 24 + N * 32 + n_obs * 56 + N * 48


def measured_flops(pkg, scene, ticks=24):
    """Exact FP64 operation count of the bench workload, from the oracle
    compiled with an instrumented scalar type (oracle/flopcount: every + - * /
    sqrt exp compare of the restatement is counted), over `ticks` ticks of the
    same scene; also the fraction of agent-steps with >= 1 in-shell obstacle
    (SURVEY.md 8d). Test infrastructure used as a measuring device only."""
    try:
        from oracle import flopcount
        return flopcount.count_scene(scene, ticks)
    except Exception as e:  # the bench line must not depend on it
        return {"error": "%s: %s" % (type(e).__name__, e)}


def cpu_baseline(config, budget_s, episode=None):
    """The CPU oracle (oracle/, a scalar C restatement of the reference's
    algorithm = kind 'port') timed on this host by oracle/cpu_bench.py in a
    process of its own: gcc -O2 and gcc -O3 -march=native (compiled here), one
    pinned core and the agents' rollouts on OpenMP threads pinned one per allowed
    physical core, >= 3 s of warm-up, up to 30 repetitions of a fixed tick count
    each: best / median / min and the share of repetitions within 10 % of the
    best (oracle/cpu_bench.py says why `best` is the number that
    reproduces on a shared host)."""
    ep = episode or (64 if config in ("C3", "task_static1") else 256)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--config", config,
                        "--budget", str(budget_s), "--episode", str(ep)], capture_output=True, text=True, timeout=120 + 6 * budget_s)
    if r.returncode != 0:
        return {"error": r.stderr[-400:]}
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    b = d["builds"]
    ok = {k: v for k, v in b.items() if "multi" in v}
    if not ok:
        return {"error": json.dumps(b)[:400]}
    best_name, best = max(ok.items(), key=lambda kv: kv[1]["multi"]["best"])
    out = {
        "value": best["multi"]["median"], "best": best["multi"]["best"], "unit": "rollouts/s", "cores": best["threads"], "kind": "port",
        "share_of_repetitions_within_10pct_of_best": best["multi"]["share_within_10pct_of_best"],
        "cpu_model": d["cpu_model"], "host_cpus": d["host_cpus"], "affinity_cpus": d["affinity_cpus"],
        "cgroup_cpu_quota": d["cgroup_cpu_quota"], "physical_cores_allowed": d["physical_cores_used"], "pinning": d["pinning"],
        "build_of_value": best_name,
        "sample": "oracle/libpmaf_oracle.so (scalar C restatement of the reference) on the bench workload %s: per build "
                  ">= %.0f s of multi-threaded warm-up, then %d repetitions of %d ticks on 1 pinned core and %d repetitions of "
                  "%d ticks with the agents' rollouts on %d OpenMP threads (one per allowed physical core, pinned; thread count = "
                  "the best repetition of a probe over %s; never more threads than agents; rest of the tick serial); value = "
                  "MEDIAN, best = fastest repetition of the better build; builds bit-identical: %s"
                  % (config, best["warmup_s"], best["one_core"]["reps"], best["one_core"]["ticks_per_rep"], best["multi"]["reps"],
                     best["multi"]["ticks_per_rep"], best["threads"], sorted(int(k) for k in best["probe_best_by_threads"]),
                     d["builds_bit_identical"]),
        "value_1core": best["one_core"]["median"], "best_1core": best["one_core"]["best"], "h_eff": best["multi"].get("h_eff"),
    }
    for tag, bb in b.items():
        if "multi" in bb:
            out["value_" + tag] = bb["multi"]["median"]
            out["best_" + tag] = bb["multi"]["best"]
            out["spread_" + tag] = [bb["multi"]["min"], bb["multi"]["max"]]
            out["share_within_10pct_" + tag] = bb["multi"]["share_within_10pct_of_best"]
            out["threads_" + tag] = bb["threads"]
            out["value_1core_" + tag] = bb["one_core"]["median"]
            out["best_1core_" + tag] = bb["one_core"]["best"]
            out["spread_1core_" + tag] = [bb["one_core"]["min"], bb["one_core"]["max"]]
        else:
            out["error_" + tag] = bb.get("error", "not built")
    return out


def cxx_boundary_latency(n_agents=64, cap=201, samples=400):
    """One planCallback through the C++ facade (tests/cpp/facade_tick lat, built by __graft_entry__.build()) on an idle
    stream, no interpreter in the way: CfManager::planTick open loop, setRealEEAgentPosition + planTick closed loop, and
    the UNCHANGED node's five individual calls (B/src/panda_bimanual_control.cpp:333-352). static1 scene."""
    import re
    v = os.environ.get("PMAF_VARIANT", "")
    exe = os.path.join(ROOT, "tests", "cpp", "facade_tick" + ("_" + v if v else ""))
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "lat", str(n_agents), str(cap), str(samples)], capture_output=True, text=True, timeout=120)
    except (OSError, subprocess.TimeoutExpired) as e:
        return {"error": str(e)}
    if r.returncode != 0:
        return {"error": r.stderr[-300:]}
    out = {"agents": n_agents, "max_prediction_steps": cap, "samples": samples, "scene": "static1 (9 + 1 obstacles)"}
    for key, what in (("plan_tick_open_loop", "planTick, open loop"), ("plan_tick_closed_loop", "setRealEEAgentPosition + planTick, closed loop"),
                      ("five_calls", "the node's five calls (stop ... start)")):
        m = re.search(re.escape(what) + r"\s+median\s+([0-9.]+)\s+p90\s+([0-9.]+)\s+p99\s+([0-9.]+)\s+max\s+([0-9.]+)", r.stdout)
        if m:
            out[key] = {"median": float(m.group(1)), "p90": float(m.group(2)), "p99": float(m.group(3)), "max": float(m.group(4))}
    return out


def kernel_name_of(cfg, n_obs, math=2):
    """math: the arithmetic policy's template argument (2 = strict default, 3 = contracted)"""
    tiles = (n_obs - 1 + 63) // 64
    if 61 <= n_obs - 1 <= 64:
        tiles = 2
    generic = os.environ.get("PMAF_FORCE_GENERIC") == "1"
    plain = os.environ.get("PMAF_PLAIN_STEP", "1")[:1] != "0"
    if cfg.get("waves_per_agent", 1) > 1:   # 62..256 obstacles, every wave with a SIMD of its own (pmaf_get_waves_per_agent)
        # <W, MATH, PLAIN, PRE>: PRE = <= 61 obstacles per wave (the riders' lanes are free)
        return "k_rollout_mw<%d, %d, %s, %s>" % (cfg["waves_per_agent"], math, "true" if plain else "false",
                                                  "true" if cfg["obstacles_per_wave"] <= 61 else "false")
    if cfg["lanes_per_agent"] == 64 and tiles <= 4 and not generic:
        t = 1 if tiles <= 1 else 2 if tiles == 2 else 4
        dpp = True    # (rounds 1-2: LDS batches for <= 20 obstacles; round 3: the DPP chain for every count)
        if os.environ.get("PMAF_SUM"):
            dpp = t > 1 or os.environ["PMAF_SUM"].startswith("d")
        if cfg.get("priority_slices") and t == 1 and dpp and plain:   # two waves per SIMD (pmaf_get_priority_slices)
            return "k_rollout_w64_sliced<%d>" % math
        # <TILES, MATH_XACT, DPPSUM, PLAIN>; the bench scenes have k_attr != 0 and unit mass = the PLAIN step
        return "k_rollout_w64<%d, %d, %s, %s>" % (t, math, "true" if dpp else "false", "true" if plain else "false")
    if cfg["lanes_per_agent"] in (8, 16, 32) and (n_obs - 2) // cfg["lanes_per_agent"] + 1 <= 4 and not generic:
        tl = (n_obs - 2) // cfg["lanes_per_agent"] + 1
        return "k_rollout_grp<%d, %d, %d>" % (cfg["lanes_per_agent"], 1 if tl <= 1 else 2 if tl == 2 else 4, math)
    return "k_rollout<%d>" % cfg["lanes_per_agent"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) under torch.distributed.run
    with the same arguments and pass their output through"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL, the peer mailboxes) -- see the environment notes
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %d ranks under torch.distributed.run\n" % (n, n))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


class Ctx:
    """what every workload of one invocation shares: ranks, process group, package"""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        # test hooks for a 1-GPU box: PMAF_BENCH_BACKEND=gloo runs torch's collectives on CPU tensors and the winner
        # exchange over a host-transport communicator, PMAF_BENCH_SINGLE_DEVICE=1 maps every rank to device 0
        self.backend = os.environ.get("PMAF_BENCH_BACKEND", "nccl")
        self.single_device = os.environ.get("PMAF_BENCH_SINGLE_DEVICE") == "1"
        if self.single_device:
            self.local_rank = 0
        self.red_dev = "cuda" if self.backend == "nccl" else "cpu"
        # PMAF_BENCH_FORCE_DIST=1: torch.distributed + an RCCL communicator even for one rank
        # -- exercises RCCL and this library's HIP runtime in one process on a 1-GPU box
        self.force_dist = os.environ.get("PMAF_BENCH_FORCE_DIST") == "1"
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a "
                             "%d-GPU line as a %d-GPU one" % (args.gpus, self.world, self.world, args.gpus))
        if self.force_dist and self.world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if self.world > 1 or self.force_dist:
            import torch                      # before libpmaf_hip.so: one HIP runtime per process
            import torch.distributed as dist
            self.dist = dist
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend)
        self.pkg = graft.load_package()
        self.pkg.load_library()
        self.n_devices = self.pkg.device_count()
        if not self.single_device and self.world > self.n_devices:
            raise SystemExit("bench.py: --gpus %d but hipGetDeviceCount() = %d (set PMAF_BENCH_SINGLE_DEVICE=1 to let "
                             "several ranks share device 0 in tests)" % (self.world, self.n_devices))
        self._groups = {}
        self.comms = {}      # exchange communicators by rank set (make_exchange_comm)

    def subgroup(self, ranks):
        """gloo side group over `ranks` (created collectively by ALL ranks, cached)"""
        key = tuple(ranks)
        if key not in self._groups:
            self._groups[key] = self.dist.new_group(ranks=list(ranks), backend="gloo")
        return self._groups[key]

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=self.red_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        if self.dist is not None:
            if self.red_dev == "cuda":
                import torch
                torch.cuda.synchronize()
            self.dist.barrier()


def make_exchange_comm(ctx, ranks):
    """the communicator of the per-tick winner-record all-gather over `ranks` (collective over ALL ranks of the job:
    the unique id / the fall-back decision travel through the process group). Returns (comm, transport) -- (None, None)
    on ranks outside `ranks`."""
    pkg, dist, world = ctx.pkg, ctx.dist, ctx.world
    n = len(ranks)
    me = ranks.index(ctx.rank) if ctx.rank in ranks else -1
    if tuple(ranks) in ctx.comms:        # one communicator per set of ranks and invocation (ncclCommInitRank is not free)
        return ctx.comms[tuple(ranks)]
    transport = "rccl" if ctx.backend == "nccl" else "host"
    comm = None
    if transport == "rccl":
        # the library's own RCCL communicator; should its bootstrap fail on any rank (environment), every rank falls
        # back to the host transport over a gloo side group so that the run still measures the exchange
        import torch
        ok = 1
        box = [pkg.PmafComm.unique_id() if ctx.rank == ranks[0] else None]
        if world > 1:
            dist.broadcast_object_list(box, src=ranks[0])
        try:
            if os.environ.get("PMAF_BENCH_FAIL_RCCL") == "1":   # test hook for the fallback below
                raise RuntimeError("PMAF_BENCH_FAIL_RCCL")
            if me >= 0:
                comm = pkg.PmafComm.rccl(n, me, box[0], ctx.local_rank)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("rank %d: RCCL communicator failed (%s)\n" % (ctx.rank, e))
            comm, ok = None, 0
        flag = torch.tensor([ok], dtype=torch.int32, device=ctx.red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                comm.close()
                comm = None
            transport = "host"
    if transport == "host":
        side = ctx.subgroup(ranks) if (n != world or ctx.backend == "nccl") else None

        class _Side:  # the few calls torch_host_allgather makes, bound to the side group
            @staticmethod
            def get_world_size():
                return n

            @staticmethod
            def all_gather_into_tensor(out, t):
                return dist.all_gather_into_tensor(out, t, group=side)
        if me >= 0:
            comm = pkg.PmafComm.host(n, me, pkg.shard.torch_host_allgather(_Side))
    ctx.comms[tuple(ranks)] = (comm, (transport if me >= 0 else None))
    return ctx.comms[tuple(ranks)]


def run_workload(ctx, spec, args, full):
    """Times one workload on the ranks it uses; every rank of the job calls this (ranks outside spec['ranks'] only
    take part in the barriers). Returns the record on rank 0 (None elsewhere).
    spec: config, mode ('replica' | 'shard' | 'c4'), populations, dynamic, total_populations, lanes_per_agent,
    exchange (bool), min_seconds, min_blocks, steps, warmup."""
    pkg, dist, world, rank = ctx.pkg, ctx.dist, ctx.world, ctx.rank
    S = pkg.scenes
    config, mode = spec["config"], spec["mode"]
    steps, warmup = spec["steps"], spec["warmup"]
    ranks = list(range(world))
    coupled = False
    host_coupled = False    # C4 across GPUs without fine-grained inboxes: set-points out of the winner table, on the host
    mailbox_failed = False  # ... or because connecting / proving the peer mailboxes failed on some rank
    scaling = "weak"
    if mode == "c4":
        arms = S.dual_arm_scenes()
        ranks = [0, 1] if world >= 2 else [0]
        mine = ([ranks.index(rank)] if rank in ranks else []) if world >= 2 else [0, 1]
        scenes = [arms[a] for a in mine]
        total_pops = 2
        scaling = "strong"
        coupled = True
        workload = "C4 dual arm"
    elif mode == "shard":
        total_pops = spec["total_populations"]
        if total_pops % world:
            raise SystemExit("--shard: --total-populations must be a multiple of the number of GPUs")
        mine = pkg.shard.partition_populations(total_pops, world, rank)
        scenes = [S.config_scene(config, scene_id=s, dynamic=spec["dynamic"]) for s in mine]
        scaling = "strong"
        workload = "%s sharded" % config
    elif config == "task_static1":
        # the reference's OWN operating point (B/config/tasks/dual_arms_static1.yaml:2,15,19): 10 agents,
        # max_prediction_steps 1500, 9 + 1 obstacles, 100 Hz -- rollouts stop early in the goal region (h_eff stated)
        scenes = [S.static1_scene(10, 1499)]
        total_pops = world
        workload = "task_static1 (dual_arms_static1.yaml as shipped)"
    else:
        # weak scaling = the SAME work on every GPU: all ranks plan the same scene(s) unless --distinct-scenes
        # (the seeded scenes differ by +-3 % in tick time, which would read as a scaling loss of the slowest one)
        P = spec["populations"]
        first = rank * P if args.distinct_scenes else 0
        mine = list(range(first, first + P))
        scenes = [S.config_scene(config, scene_id=s, dynamic=spec["dynamic"]) for s in mine]
        total_pops = P * world
        workload = config
        if spec.get("only_rank0"):
            # the WHOLE workload on rank 0's GPU while the other ranks wait at the barriers: the one-GPU reference point of a
            # strong-scaling record, measured in the same job on the same box (scaling_c5.one_gpu_same_job)
            ranks = [0]
            scenes = [S.config_scene(config, scene_id=s, dynamic=spec["dynamic"]) for s in range(P)]
            total_pops = P
            scaling = "strong"
            workload = "%s, all %d populations on ONE GPU (rank 0)" % (config, P)
    part = rank in ranks
    n_part = len(ranks)
    me = ranks.index(rank) if part else -1

    planner = comm = transport = None
    exchange = spec["exchange"] and (world > 1 or ctx.force_dist)
    if exchange:
        comm, transport = make_exchange_comm(ctx, ranks)   # collective over all ranks
    rec = {}
    if part:
        sc = scenes[0]
        N, H, n_obs = sc["n_agents"], sc["max_prediction_steps"] - 1, sc["obstacles"].shape[0]
        P = len(scenes)
        starts = np.stack([s["start"] for s in scenes])
        planner = pkg.PmafPlanner(scenes, device=ctx.local_rank, lanes_per_agent=spec["lanes_per_agent"], mgr_init_pos=starts,
                                  contracted=(spec.get("policy", "strict") == "contracted"))
        planner.set_initial_position(starts)
        obs = np.stack([s["obstacles"] for s in scenes])
        dt, cg, ws = sc["dt"], sc["cost_gains"], sc["ws_limits"]
        if comm is not None:
            planner.attach_comm(comm)
    # ---- C4: the set-points travel through the peer mailboxes (hipIpc-mapped inboxes, no collective) ----
    if coupled:
        if n_part > 1:
            box = [None] * world
            dist.all_gather_object(box, (planner.peer_export(n_part), planner.peer_info()["fine_grained"]) if part else None)
            # peers on different GPUs store into each other's inboxes while the kernels run: that needs fine-grained
            # inboxes on both sides (pmaf_peer_connect refuses otherwise). Every rank sees the same table, so the decision
            # is the same everywhere. Without them the run is NOT skipped: the arms are coupled through the host instead --
            # each tick waits for the all-gathered winner records and takes the other arm's set-point out of them
            # (shard.DualArmCoupling; bit-identical to the mailbox coupling, tests/test_shard_gpu.py
            # test_c4_dual_arm_one_arm_per_rank_hip_planner) -- and the record says so.
            no_fine = (not all(box[r][1] for r in ranks) and os.environ.get("PMAF_BENCH_SINGLE_DEVICE") != "1") \
                or os.environ.get("PMAF_BENCH_C4_HOST_COUPLED") == "1"
            if no_fine and comm is None:
                ctx.barrier()
                if part:
                    planner.close()
                return {"skipped": "C4 one arm per GPU: no fine-grained inboxes (pmaf_peer_info) and no exchange communicator "
                                   "(--no-exchange) to couple the arms through"} if rank == 0 else None
            if no_fine:
                host_coupled = True
                coupled = False
            else:
                # connect, couple and PROVE the mailboxes with three ticks before anything is timed: whatever a first
                # multi-GPU box holds in store (an IPC mapping that does not open, a peer store that never becomes
                # visible -> the in-kernel wait times out after PMAF_PEER_TIMEOUT_S), every rank learns of it in the same
                # reduction and the arms are coupled through the host instead -- a record, not a dead job
                bad = 0.0
                if part:
                    try:
                        if os.environ.get("PMAF_BENCH_FAIL_PEER") == "connect":   # test hook
                            raise pkg.PmafError(-2, "PMAF_BENCH_FAIL_PEER=connect")
                        planner.peer_connect(n_part, me, [box[r][0] for r in ranks])
                        pkg.shard.couple_dual_arm_on_device(planner, n_part, me, np.stack([a["start"] for a in S.dual_arm_scenes()]))
                    except pkg.PmafError as e:
                        bad = 1.0
                        sys.stderr.write("rank %d: peer mailboxes could not be connected (%s)\n" % (rank, e))
                bad = ctx.max_over_ranks(bad)
                if not bad:
                    if part:
                        try:
                            if os.environ.get("PMAF_BENCH_FAIL_PEER") == "probe":   # test hook
                                raise pkg.PmafError(-2, "PMAF_BENCH_FAIL_PEER=probe")
                            planner.tick(obs, dt, cg, ws)
                            planner.tick(None, dt, cg, ws)
                            planner.tick(None, dt, cg, ws)
                            planner.stop()
                        except pkg.PmafError as e:
                            bad = 1.0
                            sys.stderr.write("rank %d: peer mailboxes failed their probe ticks (%s)\n" % (rank, e))
                    bad = ctx.max_over_ranks(bad)
                if bad:
                    if part:
                        try:
                            planner.stop()
                        except pkg.PmafError:
                            pass
                    ctx.barrier()                 # nobody unmaps an inbox a peer may still store into
                    if part:
                        try:
                            planner.peer_disconnect()
                        except pkg.PmafError:
                            pass
                    coupled = False
                    if comm is None:
                        if part:
                            planner.close()
                        return {"skipped": "C4 one arm per GPU: the peer mailboxes failed and there is no exchange communicator "
                                           "(--no-exchange) to couple the arms through"} if rank == 0 else None
                    host_coupled = True
                    mailbox_failed = True
                if part:
                    planner.set_initial_position(starts)     # (the probe ticks moved the arms)
        elif part:
            planner.peer_connect(1, 0, [planner.peer_export(1)])
            pkg.shard.couple_dual_arm_on_device(planner, n_part, me, np.stack([a["start"] for a in S.dual_arm_scenes()]))
    hc = None
    if host_coupled and part:
        arms_all = S.dual_arm_scenes()
        hc = {"coupling": pkg.shard.DualArmCoupling(np.stack([a["obstacles"] for a in arms_all]), 0.1),
              "pos": np.stack([a["start"] for a in arms_all])}
    ctx.barrier()

    tick_no = [0]
    # C3's 500-step rollouts (1 m of travel) reach the goal region once the real agent has advanced ~0.2 m: shorter
    # episodes keep every timed rollout at its full horizon there
    episode = min(args.episode, 64) if (args.episode and config in ("C3", "task_static1")) else args.episode

    def maybe_restart_episode():
        # stationary workload: restart the episode before the real agent gets so close to the goal that rollouts stop
        # early (cf_agent.cpp:310). Inside the timed blocks (it is part of the work), but OUTSIDE the per-tick latency
        # samples: set_initial_position waits for the running rollout and launches a kernel of its own, which is not
        # what "latency of a tick" means (it was the p99 of the round-2 line: one restart among 100 samples)
        if episode and tick_no[0] % episode == 0:
            planner.set_initial_position(starts)
            if hc is not None:
                hc["pos"] = np.stack([a["start"] for a in S.dual_arm_scenes()])

    def one_tick(o):
        tick_no[0] += 1
        if hc is not None:   # host-coupled C4: this arm's trailing obstacle = the other arm's set-point of the last tick
            lo = hc["coupling"].coupled_obstacles(hc["pos"])
            b = planner.tick(lo[me], dt, cg, ws)
            hc["pos"] = planner.winners_wait()[:, 0, 4:7].copy()   # (the collective is ON the control path here)
            return b
        return planner.tick(o if spec["dynamic"] else None, dt, cg, ws)

    def sync_all():
        if part:
            planner.stop()
            if comm is not None:
                planner.winners_wait()
        ctx.barrier()

    if part:
        obs0 = obs.copy()
        if hc is not None:
            one_tick(obs)
        else:
            planner.tick(None if coupled else obs, dt, cg, ws)  # obstacles resident in HBM from here on
            tick_no[0] += 1
        for _ in range(warmup):
            maybe_restart_episode()
            one_tick(obs)
        # HIP events on every `--time-every`-th rollout launch of the timed region (on the kernel's own dispatch packet):
        # a timed dispatch costs 3-5 us of device time per tick (profiles/r4_tick_overhead.txt), so the average kernel
        # duration is taken from a sample of the launches instead of slowing every tick down
        planner.set_profiling(max(1, args.time_every))
    sync_all()
    if part:
        planner.reset_kernel_stats()
        planner.exchange_times_us()  # clear
        if coupled:
            planner.peer_times_us()

    # ---- timed region: blocks of `steps` ticks, barrier + device sync on both sides of each ----
    block_s, lat = [], []
    total_timed, n_blocks, max_blocks = 0.0, 0, 10000
    while True:
        blk_lat = np.zeros(steps)
        t0 = time.perf_counter()
        if part:
            for k in range(steps):
                maybe_restart_episode()
                ta = time.perf_counter()
                one_tick(obs)
                blk_lat[k] = time.perf_counter() - ta
                if spec["dynamic"]:
                    # moving obstacles: advanced like dynamic_obstacle_node does, put back with the agent at every
                    # episode start so the workload stays stationary (they would drift out of the scene otherwise)
                    if episode and tick_no[0] % episode == 0:
                        obs = obs0.copy()
                    else:
                        obs = np.stack([S.advance_live_obstacles(o) for o in obs])
            planner.stop()
            if comm is not None:
                planner.winners_wait()
            if dist is not None and ctx.red_dev == "cuda":
                import torch
                torch.cuda.synchronize()
        el = ctx.max_over_ranks((time.perf_counter() - t0) if part else 0.0)   # the all-reduce closes the block
        block_s.append(el)
        lat.append(blk_lat)
        total_timed += el
        n_blocks += 1
        # same decision on every rank (el is reduced). At least min_blocks blocks, so that the median is one of
        # several blocks and a single slow block (a clock or scheduling hiccup on the box) cannot move it
        if (total_timed >= spec["min_seconds"] and n_blocks >= spec["min_blocks"]) or n_blocks >= max_blocks:
            break
        sync_all()
    elapsed = float(np.median(block_s))
    mine_rec = None
    if part:
        lat = np.concatenate(lat)
        kernel_ms, launches, agent_steps = planner.kernel_stats()     # (launches = the TIMED ones)
        all_launches = planner.launch_count()
        cfg = planner.launch_config()
        ag_us = planner.exchange_times_us() if comm is not None else np.zeros(0)
        pw, pp = planner.peer_times_us() if coupled else (np.zeros(0), np.zeros(0))
        # set-point latency of a tick issued on an idle stream (the previous rollout has finished, as in a 100 Hz
        # control loop): host call -> best index and next set-point on the host. Outside the timed region.
        idle = np.zeros(0)
        idle_lib = (np.zeros(0), np.zeros(0))
        closed_sp = np.zeros(0)
        if full and not ((coupled or host_coupled) and n_part > 1):
            idle = np.zeros(500)
            planner.tick_times_us()          # (clears the library's own record of the timed ticks)
            for k in range(idle.size):
                maybe_restart_episode()
                planner.stop()
                ta = time.perf_counter()
                one_tick(obs)
                idle[k] = time.perf_counter() - ta
            planner.stop()
            # the same calls on the library's own clock (pmaf_get_tick_times_us): without this script's ctypes / numpy
            # / interpreter time, which is where the tail of the figure above comes from
            idle_lib = planner.tick_times_us(idle.size)
            # closed loop (open_loop: false in the task file, B/src/panda_bimanual_control.cpp:333-335): the measured position
            # -- trailing the set-point by 30 % of the last step -- handed over in front of every tick; library clock of the tick
            closed_sp = np.zeros(0)
            if not coupled and not host_coupled:
                meas = np.asarray(planner.real_state()[0], dtype=np.float64).reshape(P, 3).copy()
                for k in range(300):
                    if episode and tick_no[0] % episode == 0:
                        planner.set_initial_position(starts)
                        meas = starts.copy()
                    planner.stop()
                    planner.set_real_position(meas)
                    one_tick(obs)
                    spn = np.asarray(planner.real_state()[0], dtype=np.float64).reshape(P, 3)
                    meas = spn - 0.3 * (spn - meas)
                planner.stop()
                closed_sp = planner.tick_times_us(300)[1]
                planner.set_initial_position(starts)
            # the tick SURVEY.md 8(d) defines: host call -> costs + best index + the WINNING TRAJECTORY on the host (what the
            # reference's planCallback hands out, B/src/panda_bimanual_control.cpp:340-347). The manager kernel writes the
            # selected agent's scored path into mapped pinned memory behind the set-point (pmaf_enable_winner_path);
            # library clock: entry of pmaf_tick -> pmaf_view_winner_path returns. Idle stream, >= 1000 ticks.
            planner.enable_winner_path(True)
            wp_wall = np.zeros(1000)
            for k in range(wp_wall.size):
                maybe_restart_episode()
                planner.stop()
                if k == 0:
                    planner.winner_path_times_us()   # clear
                ta = time.perf_counter()
                one_tick(obs)
                planner.winner_path_wait()
                wp_wall[k] = time.perf_counter() - ta
            planner.stop()
            wp_lib = planner.winner_path_times_us()
            planner.enable_winner_path(False)
        mine_rec = dict(tick_us=float(np.median(lat) * 1e6), tick_p99=float(np.percentile(lat, 99) * 1e6),
                        ag_us=float(np.median(ag_us)) if ag_us.size else None,
                        ag_p99=float(np.percentile(ag_us, 99)) if ag_us.size else None, ag_n=int(ag_us.size),
                        kernel_us=kernel_ms / max(launches, 1) * 1e3, steps_per_launch=agent_steps / max(all_launches, 1),
                        timed_launches=int(launches), launches=int(all_launches),
                        peer_wait=[float(np.median(pw)), float(np.percentile(pw, 99))] if pw.size else None,
                        peer_pub=float(np.median(pp)) if pp.size else None, peer_n=int(pw.size),
                        collective_world=comm.world if comm is not None else None)
    per_rank = [mine_rec]
    if dist is not None and world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_rec)
    ctx.barrier()                        # nobody unmaps an inbox a peer may still store into
    if part:
        if coupled:
            planner.peer_disconnect()
        if comm is not None:
            planner.attach_comm(None)     # (the communicator itself is kept for the next workload on the same ranks)
        planner.close()
    if rank != 0:
        return None

    used = [r for r in per_rank if r is not None]
    r0 = used[0]
    rollouts = N * total_pops * steps
    value = rollouts / elapsed
    avg_kernel_s = r0["kernel_us"] * 1e-6
    bytes_per_launch = P * algorithmic_bytes_per_tick(N, H, n_obs)
    achieved = bytes_per_launch / avg_kernel_s / 1e9
    steps_per_launch = r0["steps_per_launch"]
    kernel_name = kernel_name_of(cfg, n_obs, 3 if spec.get("policy") == "contracted" else 2)
    rec = {
        "rollouts_per_s": value, "ms_per_tick": elapsed / steps * 1e3, "scaling": scaling, "gpus_used": n_part,
        "workload": "%s: %d agents x %d-step horizon, %d sphere obstacles + repulsive sentinel, %d population(s) per GPU "
                    "(%d in the job) on %d GPU(s), %s obstacles, one pmaf_tick per step%s%s"
                    % (workload, N, H, n_obs - 1, P, total_pops, n_part, "moving" if spec["dynamic"] else "static",
                       ", winner records all-gathered once per tick (%s)" % ("RCCL" if transport == "rccl" else "host transport")
                       if comm is not None else "",
                       ", set-points through the peer mailboxes (%s)" % ("hipIpc-mapped inboxes of the two ranks" if n_part > 1
                                                                         else "the handle's own inbox") if coupled else
                       ", arms coupled THROUGH THE HOST: every tick waits for the all-gathered winner records (fine-grained "
                       "inboxes not exportable on this runtime, or PMAF_BENCH_C4_HOST_COUPLED=1)" if host_coupled else ""),
        "agents": N, "horizon": H, "obstacles": n_obs - 1, "populations_per_gpu": P, "populations_total": total_pops,
        "h_eff": steps_per_launch / (N * P),
        "agent_steps_per_s": sum(u["steps_per_launch"] for u in used) * steps / elapsed,
        "blocks": n_blocks, "block_ticks": steps, "timed_s": total_timed,
        "block_ms": {"median": elapsed * 1e3, "min": float(np.min(block_s)) * 1e3, "max": float(np.max(block_s)) * 1e3},
        "tick_latency_us": {"median": r0["tick_us"], "p99": r0["tick_p99"], "per_rank_median": [u["tick_us"] for u in used],
                            "note": "back-to-back ticks: each call waits for the previous rollout"},
        "allgather_us": None if comm is None else {
            "median": r0["ag_us"], "p99": r0["ag_p99"], "n": r0["ag_n"], "per_rank_median": [u["ag_us"] for u in used],
            "transport": transport, "collective_world": r0["collective_world"],
            "note": ("device time between the events around ncclAllGather on the exchange stream (includes the "
                     "wait for the slowest rank); off the rollout's critical path") if transport == "rccl" else
                    "host transport: wall time of the all-gather callback (gloo), run when the table is asked for"},
        "coupling": ("peer mailboxes" if coupled else
                     ("host, winner records (peer mailboxes failed to connect / their probe ticks failed: see stderr)" if mailbox_failed
                      else "host, winner records") if host_coupled else None),
        "header_exchange_us": None if not coupled else {
            "wait_median": r0["peer_wait"][0] if r0["peer_wait"] else None,
            "wait_p99": r0["peer_wait"][1] if r0["peer_wait"] else None,
            "publish_median": r0["peer_pub"], "n": r0["peer_n"],
            "per_rank_wait_median": [u["peer_wait"][0] if u["peer_wait"] else None for u in used],
            "note": "peer mailboxes: wait = time the manager kernel of a tick spent waiting for the other arm's header of "
                    "the previous tick (all the coupling costs the control path), publish = its stores into the peers' "
                    "inboxes incl. the system-scope fence; device clock; no winners_wait on the tick path"},
        "arithmetic_policy": spec.get("policy", "strict"),
        # strict: bit-identical to the CPU oracle on this workload (tests/test_parity_gpu.py). contracted: tolerance parity
        # (selected trajectory <= 1e-5 m, same best-index sequence over >= 50 closed-loop ticks) holds on C1-C4 and does
        # NOT on C5 (scene 1: 3.2e-3 m) -- tests/test_tolerance_gpu.py CONTRACTED_EXCEEDS
        "parity": ("bit-exact vs oracle" if spec.get("policy", "strict") == "strict" else "tolerance 1e-5 m on the selected trajectory"),
        "parity_met": (True if spec.get("policy", "strict") == "strict" else config != "C5"),
        "kernel": kernel_name, "avg_kernel_us": r0["kernel_us"], "per_rank_kernel_us": [u["kernel_us"] for u in used],
        "kernel_timing": {"launches_in_timed_region": r0["launches"], "launches_timed_with_hip_events": r0["timed_launches"],
                          "every": max(1, args.time_every)},
        "lanes_per_agent": cfg["lanes_per_agent"], "rollout_blocks": cfg["n_blocks"],
        "algorithmic_bytes_per_launch": bytes_per_launch, "hbm_achieved_gbs": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
    }
    if config == "task_static1":
        period_ms = 1e3 * sc["dt"]
        rec["regime"] = {
            "tick_budget_ms": period_ms, "rollout_launch_us": r0["kernel_us"], "back_to_back_tick_us": elapsed / steps * 1e6,
            "share_of_the_control_period": r0["kernel_us"] * 1e-3 / period_ms,
            "us_per_step_of_the_longest_chain": r0["kernel_us"] / H,
            "note": "the reference's shipped operating point: 10 agents x up to %d steps x %d + 1 obstacles at %.0f Hz. A "
                    "launch lasts as long as its longest chain (full horizon at the episode's start, shorter as the real "
                    "agent nears the goal: h_eff); few long chains are the shape where one GPU wave per agent (one issue slot "
                    "per 4 cycles) is SLOWER per rollout than one x86 core per agent -- see cpu_port beside this record; both "
                    "fit the control period several times over, and the set-point latency (setpoint_latency_us of the "
                    "headline) does not depend on the horizon" % (H, n_obs - 1, 1e3 / period_ms)}
    if full:
        rec["_wp"] = (wp_lib, wp_wall) if (full and part and not ((coupled or host_coupled) and n_part > 1)) else (np.zeros(0), np.zeros(0))
        rec["_idle"] = idle
        rec["_idle_lib"] = idle_lib
        rec["_closed_sp"] = closed_sp if (part and not ((coupled or host_coupled) and n_part > 1)) else np.zeros(0)
        rec["_scene"] = sc
        rec["_P"] = P
        rec["_transport"] = transport
        rec["_has_comm"] = comm is not None
    return rec


C5_SCALING_REASON = (
    "BASELINE C5 = 8 independent scenes x 1024 agents x 200 steps x 32 obstacles, scene s on GPU s mod N (agents are "
    "independent, B/src/cf_manager.cpp:118-123: population per GPU, no collective on the rollout's data path). A rollout is ONE "
    "dependent 200-step chain; a launch cannot be shorter than its longest chain at one wave per SIMD (~235 us for 1024 agents = "
    "1024 waves on 1024 SIMDs), however few scenes a GPU holds. One GPU runs all 8 scenes in ~0.7 ms (16 lanes per agent, two "
    "waves per SIMD), so 8 GPUs can gain at most ~0.7 / 0.235 = 3x: strong scaling of this configuration saturates by "
    "construction, it is not a communication loss (the winner-record all-gather runs on a side stream beside the next rollout).")


def emulate_c5_scaling(ctx, args, block_ticks=20, min_seconds=0.15, min_blocks=3):
    """BASELINE C5's strong-scaling curve, emulated on ONE GPU: for N = 1, 2, 4, 8 every rank's share of the eight scenes
    (pkg.shard.partition_populations: scene s on rank s mod N -- what `--shard` deals out on N GPUs) is run in a handle of
    its own, one after the other, with bench.py's timing rules (warm-up, barrier-free here: one process; blocks of ticks,
    median block; episodes restarted like the timed workloads). The job's tick at N GPUs = the SLOWEST rank's tick (the
    ranks run concurrently on a real node and the line's time is the max over ranks); the winner-record all-gather is
    not on the tick's critical path (side stream, two slots: a tick never waits for the previous tick's collective).
    Returns {"predicted_by_n": {N: {...}}, ...}. world == 1 only."""
    pkg = ctx.pkg
    S = pkg.scenes
    by_n = {}
    episode = args.episode
    for n in (1, 2, 4, 8):
        ranks = []
        for r in range(n):
            mine = pkg.shard.partition_populations(8, n, r)
            scenes = [S.config_scene("C5", scene_id=s) for s in mine]
            sc = scenes[0]
            starts = np.stack([q["start"] for q in scenes])
            obs = np.stack([q["obstacles"] for q in scenes])
            planner = pkg.PmafPlanner(scenes, device=ctx.local_rank, mgr_init_pos=starts)
            planner.set_initial_position(starts)
            dt, cg, ws = sc["dt"], sc["cost_gains"], sc["ws_limits"]
            planner.tick(obs, dt, cg, ws)
            tick_no = 1
            for _ in range(10):
                planner.tick(None, dt, cg, ws)
                tick_no += 1
            planner.set_profiling(max(1, args.time_every))
            planner.stop()
            planner.reset_kernel_stats()
            blocks, timed = [], 0.0
            while timed < min_seconds or len(blocks) < min_blocks:
                t0 = time.perf_counter()
                for _ in range(block_ticks):
                    if episode and tick_no % episode == 0:
                        planner.set_initial_position(starts)
                    planner.tick(None, dt, cg, ws)
                    tick_no += 1
                planner.stop()
                el = time.perf_counter() - t0
                blocks.append(el)
                timed += el
            kernel_ms, launches, agent_steps = planner.kernel_stats()
            all_launches = planner.launch_count()
            cfg = planner.launch_config()
            planner.close()
            ranks.append({"rank": r, "scenes": [int(x) for x in mine], "ms_per_tick": float(np.median(blocks)) / block_ticks * 1e3,
                          "kernel_us": kernel_ms / max(launches, 1) * 1e3, "lanes_per_agent": cfg["lanes_per_agent"],
                          "h_eff": agent_steps / max(all_launches, 1) / (sc["n_agents"] * len(scenes))})
        ms = max(x["ms_per_tick"] for x in ranks)
        by_n[n] = {"populations_per_gpu": 8 // n, "ms_per_tick": ms, "rollouts_per_s": 8 * 1024 / (ms * 1e-3),
                   "kernel_us_slowest_rank": max(x["kernel_us"] for x in ranks),
                   "lanes_per_agent": ranks[0]["lanes_per_agent"],
                   "per_rank_ms_per_tick": [x["ms_per_tick"] for x in ranks],
                   "h_eff_min": min(x["h_eff"] for x in ranks)}
    t1 = by_n[1]["ms_per_tick"]
    for n, row in by_n.items():
        row["speedup_vs_1gpu"] = t1 / row["ms_per_tick"]
        row["efficiency_vs_1gpu"] = t1 / row["ms_per_tick"] / n
    return {"method": "emulated on ONE GPU: each rank's share of the 8 scenes in a handle of its own, sequentially; tick at N GPUs = "
                      "the slowest rank's; all-gather off the critical path (side stream)",
            "predicted_by_n": {str(n): by_n[n] for n in sorted(by_n)},
            "chain_floor_us": by_n[8]["kernel_us_slowest_rank"],
            "reason": C5_SCALING_REASON}


def committed_c5_prediction():
    """the emulated table a one-GPU run left in profiles/ (tools/r6_evidence.sh), for N > 1 lines to be read against"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r6_scaling_emulated.json")))
        return {"source": "profiles/r6_scaling_emulated.json (one MI355X, round 6)", "predicted_by_n": d["predicted_by_n"],
                "allgather": d.get("allgather")}
    except (OSError, ValueError, KeyError):
        return None


# per-tick estimates (ms, one MI355X, round-4 measurements) the launch plan's time budget is computed from
EST_TICK_MS = {"C1": 0.125, "C2": 0.245, "C3": 1.03, "C4": 0.30, "C5": 0.75, "task_static1": 1.8}
C5_EMULATION_EST_S = 8.0   # emulate_c5_scaling: 15 handles, >= 0.15 s timed each + set-up


def build_plan(args, world):
    """The workloads one invocation times: (headline spec, [(name, sub-configuration spec)], {name: {"skipped": why}}).
    The same function serves N = 1 and N > 1 (SCALE's N = 1 point and the single-GPU BENCH line are the same code
    path); what depends on the world size is only which ranks take part in C4 (two) and how C5's eight scenes are
    dealt out (scene s on rank s mod N; skipped when 8 is not a multiple of N)."""
    default_headline = (args.config == "C2" and not args.shard and args.populations == 1 and not args.dynamic
                        and args.lanes_per_agent == 0)
    head_spec = dict(config=args.config, mode="c4" if args.config == "C4" else "shard" if args.shard else "replica",
                     populations=args.populations, dynamic=args.dynamic, total_populations=args.total_populations,
                     lanes_per_agent=args.lanes_per_agent, exchange=not args.no_exchange, steps=args.steps,
                     warmup=args.warmup, min_seconds=args.min_seconds, min_blocks=args.min_blocks)
    plan, skipped = [], {}
    if default_headline and not args.only_headline:
        sst = args.sub_steps or min(args.steps, 50)
        base = dict(populations=1, dynamic=False, total_populations=8, lanes_per_agent=0, exchange=True, steps=sst,
                    warmup=min(args.warmup, 10), min_seconds=args.sub_seconds, min_blocks=max(5, args.min_blocks))
        plan = [("C1", dict(base, config="C1", mode="replica")),
                ("C3", dict(base, config="C3", mode="replica"))]
        if 8 % world == 0:
            plan.append(("C5_sharded", dict(base, config="C5", mode="shard")))
            if world > 1:   # ... and all eight scenes on rank 0's GPU: the efficiency of scaling_c5 without a second job
                plan.append(("C5_one_gpu", dict(base, config="C5", mode="replica", populations=8, only_rank0=True, exchange=False)))
        plan.append(("C4", dict(base, config="C4", mode="c4")))
        # the regime statement: the shipped task's size, where a GPU has the least to offer -- ten 1500-step
        # chains; reported with the 10 ms budget of the 100 Hz loop and (N = 1) the CPU port at one thread per agent
        plan.append(("task_static1", dict(base, config="task_static1", mode="replica", steps=min(sst, 20), exchange=False)))
        # the opt-in contracted arithmetic policy (PMAF_FLAG_CONTRACTED: rcp / rsq sequences + FMA contraction; NOT
        # bit-exact, tolerance parity where tests/test_tolerance_gpu.py says it holds) beside the strict sub-records:
        # what the north star's 1e-5 m budget buys in kernel time. Never the headline `value`.
        plan.append(("C2_contracted", dict(base, config="C2", mode="replica", policy="contracted")))
        plan.append(("C3_contracted", dict(base, config="C3", mode="replica", policy="contracted")))
        if 8 % world == 0:
            plan.append(("C5_sharded_contracted", dict(base, config="C5", mode="shard", policy="contracted")))
        else:
            skipped["C5_sharded"] = {"skipped": "8 scenes do not divide over %d GPUs" % world}
    return head_spec, plan, skipped


def plan_budget_s(args, world):
    """Upper estimate of the GPU-side seconds of one invocation, by construction of the plan: per workload the timed
    blocks (at least min_blocks blocks and min_seconds), its warm-up, one barrier-bracketed sync per block and 0.5 s of
    set-up; for the headline the 500 + 1000 idle-stream latency samples; the CPU baseline and the flop count (N = 1 /
    rank 0 only). Interpreter start, `import torch` and the rendezvous are not in it."""
    head, plan, _ = build_plan(args, world)
    total, rows = 0.0, []
    for name, sp in [("headline", head)] + plan:
        tick = EST_TICK_MS[sp["config"]] * 1e-3
        if sp["mode"] == "shard" and sp["config"] == "C5":   # fewer scenes per GPU: not below the one-scene launch
            tick *= max(0.35, sp["total_populations"] / float(world) / 8.0)
        block = sp["steps"] * tick
        timed = max(sp["min_seconds"] + block, sp["min_blocks"] * block)
        t = 0.5 + sp["warmup"] * tick + timed + 0.002 * max(sp["min_blocks"], timed / max(block, 1e-9))
        if name == "headline":
            t += 1500 * (tick + 60e-6)
        rows.append((name, t))
        total += t
    if world == 1:
        # both oracle builds (budget + >= 3 s of warm-up each), then the task_static1 CPU port (<= 8 s + warm-up)
        total += (args.cpu_seconds + 6.0 + 2.0) + ((min(args.cpu_seconds, 8.0) + 6.0 + 2.0) if plan else 0.0)
    total += 2.0 if args.flop_ticks > 0 else 0.0
    if world == 1 and plan:
        total += C5_EMULATION_EST_S
        rows.append(("C5 scaling emulation", C5_EMULATION_EST_S))
    return total, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="C2", help="C1|C2|C3|C4|C5 (BASELINE.json configs); C2 is the metric's config")
    ap.add_argument("--populations", type=int, default=1, help="independent populations per GPU in one handle")
    ap.add_argument("--shard", action="store_true",
                    help="strong scaling: --total-populations scenes (default 8 = BASELINE C5) partitioned over the "
                         "ranks, scene s on rank s %% N")
    ap.add_argument("--total-populations", type=int, default=8)
    ap.add_argument("--no-exchange", action="store_true", help="N > 1: no winner-record all-gather (independent replicas)")
    ap.add_argument("--lanes-per-agent", type=int, default=0)
    ap.add_argument("--dynamic", action="store_true", help="moving obstacles, re-uploaded every tick")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget, warm-up not included (0 = skip)")
    ap.add_argument("--flop-ticks", type=int, default=24, help="ticks of the instrumented-oracle flop count (0 = skip)")
    ap.add_argument("--min-blocks", type=int, default=5, help="time at least this many blocks of --steps ticks")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat blocks of --steps ticks until this much has been timed; the median block is reported")
    ap.add_argument("--distinct-scenes", action="store_true",
                    help="N > 1, weak scaling: rank r plans scene(s) r*P .. r*P+P-1 instead of every rank planning the "
                         "same scene(s)")
    ap.add_argument("--episode", type=int, default=256,
                    help="ticks per episode: the real agent is put back at the start every EPISODE ticks so every "
                         "timed rollout runs its full horizon (stationary workload; 0 = never)")
    ap.add_argument("--only-headline", action="store_true", help="skip the C1 / C3 / C5-sharded / C4 sub-configurations")
    ap.add_argument("--sub-steps", type=int, default=0, help="ticks per block of the sub-configurations (default min(steps, 50))")
    ap.add_argument("--sub-seconds", type=float, default=0.25, help="minimum timed seconds per sub-configuration")
    ap.add_argument("--time-every", type=int, default=8,
                    help="HIP-event timing on every n-th rollout launch of the timed region (1 = every launch)")
    ap.add_argument("--no-c5-emulation", action="store_true",
                    help="N = 1: skip the emulated 1/2/4/8-GPU curve of BASELINE C5 (scaling_c5.prediction, ~8 s)")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, build the process group and the exchange communicator, all-gather through it "
                         "once and print the launch plan -- no planner, no GPU work (CPU test of the multi-rank plumbing)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    ctx = Ctx(args)
    rank, world, dist = ctx.rank, ctx.world, ctx.dist
    if args.dry_run:
        who = [(rank, ctx.local_rank, os.getpid())]
        got = None
        if dist is not None:
            who = [None] * world
            dist.all_gather_object(who, (rank, ctx.local_rank, os.getpid()))
            comm, transport = make_exchange_comm(ctx, list(range(world)))
            got = comm.allgather(np.array([float(rank)])).reshape(-1).tolist()
            collective_world = comm.world
            comm.close()
            ctx.comms.clear()
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            budget, rows = plan_budget_s(args, world)
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": who, "devices_visible": ctx.n_devices,
                              "plan": [n for n, _ in build_plan(args, world)[1]], "plan_skipped": sorted(build_plan(args, world)[2]),
                              "headline": build_plan(args, world)[0],
                              # how C4's arms will be coupled on this launch (run_workload, mode "c4"): decided at run time from
                              # pmaf_peer_info of every rank -- never a skipped record while an exchange communicator exists
                              "c4_coupling": ("one handle: the handle's own inbox" if world == 1 else
                                              "host, winner records (forced: PMAF_BENCH_C4_HOST_COUPLED=1)"
                                              if os.environ.get("PMAF_BENCH_C4_HOST_COUPLED") == "1" else
                                              "peer mailboxes between ranks 0 / 1; host-coupled through the winner records if an "
                                              "inbox is not fine-grained device memory"),
                              "budget_s": budget, "budget_rows": rows,
                              "transport": transport if got is not None else None,
                              "collective_world": collective_world if got is not None else None,
                              "allgather_of_ranks": got}), flush=True)
        return
    head_spec, plan, skipped = build_plan(args, world)
    head = run_workload(ctx, head_spec, args, full=True)

    subs = {}
    for name, spec in plan:
        r = run_workload(ctx, spec, args, full=False)
        if rank == 0:
            subs[name] = r
    if rank == 0:
        subs.update(skipped)
    emu = None
    if world == 1 and plan and not args.no_c5_emulation:
        emu = emulate_c5_scaling(ctx, args)

    line = None
    if rank == 0 and "skipped" in head:      # (--config C4 on GPUs whose runtime cannot export fine-grained inboxes)
        line = json.dumps({"metric": "agent_rollouts_per_s", "value": None, "skipped": head["skipped"], "n_gpus": world})
    elif rank == 0:
        idle, sc, P = head.pop("_idle"), head.pop("_scene"), head.pop("_P")
        wp_lib, wp_wall = head.pop("_wp")
        idle_enq, idle_sp = head.pop("_idle_lib")
        closed_sp = head.pop("_closed_sp")
        transport, has_comm = head.pop("_transport"), head.pop("_has_comm")
        N, H, n_obs = head["agents"], head["horizon"], head["obstacles"] + 1
        kernel_name, avg_kernel_s = head["kernel"], head["avg_kernel_us"] * 1e-6
        steps_per_launch = head["h_eff"] * N * P
        # exact flop count of this workload from the instrumented oracle
        fl = measured_flops(ctx.pkg, sc, args.flop_ticks) if args.flop_ticks > 0 else {"error": "skipped"}
        if "flops_per_agent_step" in fl:
            flops_step, flop_src = fl["flops_per_agent_step"], "measured"
        else:
            M = n_obs - 1   # SURVEY 8(d) estimate as the fall-back
            flops_step, flop_src = 110 + 47 * M + 12 * M + 65 * (M / 4.0), "SURVEY 8(d) estimate (flop count unavailable)"
        flops = flops_step * steps_per_launch
        # HBM traffic per launch of this kernel from the committed PMC passes (tools/gpu_prof.sh +
        # tools/traffic_from_pmc.py); PMC counters cannot be collected from inside the timed run itself
        traffic = traffic_src = pmc = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            key = "%s%s:%s" % (args.config, "" if P == 1 else "x%d" % P, kernel_name)
            if key in tj and not args.dynamic:
                traffic = tj[key]["traffic_bytes_per_launch"]
                traffic_src = "profiles/traffic.json (rocprofv3 --pmc passes of this command, NOT this run): " + \
                              tj[key].get("source", "")
                pmc = {k: tj[key].get(k) for k in ("valu_active_frac_of_wave_cycles", "valu_insts_per_launch", "waves_per_launch")}
        except (OSError, ValueError, KeyError):
            traffic = None
        out = {
            "metric": "agent_rollouts_per_s", "value": head["rollouts_per_s"], "unit": "rollouts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_tick"], "higher_is_better": True,
            "scaling": head["scaling"], "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": head["workload"],
                       "agents": N, "horizon": H, "obstacles": n_obs - 1, "populations_per_gpu": P,
                       "populations_total": head["populations_total"],
                       "parallelism": "population-per-gpu x%d (%s)" % (world, "sharded scenes" if head["scaling"] == "strong" else
                                                                      "distinct scenes" if args.distinct_scenes else
                                                                      "same scene on every GPU"),
                       "collective": None if not has_comm else
                       "%s of %d B winner records per rank per tick, second stream, overlapped with the "
                       "rollout" % ("ncclAllGather" if transport == "rccl" else "host-transport all-gather",
                                    P * (8 + 3 * (H + 1)) * 8),
                       "collective_world": head["allgather_us"]["collective_world"] if head["allgather_us"] else None,
                       "devices_visible": ctx.n_devices,
                       "lanes_per_agent": head["lanes_per_agent"], "rollout_blocks": head["rollout_blocks"],
                       "arithmetic": "f64, hand-expanded IEEE div/sqrt sequences (default policy; bit-identical to "
                                     "the CPU oracle)"},
            "timing": {"blocks": head["blocks"], "block_ticks": args.steps, "timed_s": head["timed_s"],
                       "block_ms": head["block_ms"],
                       "note": "value / ms_per_step = MEDIAN block of `steps` ticks (max over ranks per block)"},
            "agent_steps_per_s": head["agent_steps_per_s"],
            "h_eff": head["h_eff"],
            "tick_latency_us": head["tick_latency_us"],
            "setpoint_latency_us": None if not idle.size else {
                "median": float(np.median(idle) * 1e6), "p90": float(np.percentile(idle, 90) * 1e6),
                "p99": float(np.percentile(idle, 99) * 1e6), "max": float(np.max(idle) * 1e6), "n": int(idle.size),
                "in_library": None if not idle_sp.size else {
                    "median": float(np.median(idle_sp)), "p90": float(np.percentile(idle_sp, 90)),
                    "p99": float(np.percentile(idle_sp, 99)), "max": float(np.max(idle_sp)),
                    "enqueue_median": float(np.median(idle_enq)), "enqueue_p99": float(np.percentile(idle_enq, 99)),
                    "note": "the same calls on the library's own clock (pmaf_get_tick_times_us): entry of pmaf_tick -> "
                            "set-point on the host; enqueue = both launches handed to the stream"},
                "closed_loop_in_library": None if not closed_sp.size else {
                    "median": float(np.median(closed_sp)), "p90": float(np.percentile(closed_sp, 90)),
                    "p99": float(np.percentile(closed_sp, 99)), "n": int(closed_sp.size),
                    "note": "pmaf_set_real_position(measured) in front of every pmaf_tick (the node with open_loop: false): the "
                            "position rides in pinned memory into the manager kernel -- no stream sync, no copy command"},
                "note": "tick issued on an idle stream: host call -> best index + next set-point "
                        "on the host (the new rollout then runs asynchronously); measured around the ctypes call"},
            "tick_with_winner_path_us": None if not wp_lib.size else {
                "median": float(np.median(wp_lib)), "p90": float(np.percentile(wp_lib, 90)),
                "p99": float(np.percentile(wp_lib, 99)), "max": float(np.max(wp_lib)), "n": int(wp_lib.size),
                "around_the_ctypes_calls": {"median": float(np.median(wp_wall) * 1e6), "p99": float(np.percentile(wp_wall, 99) * 1e6)},
                "path_bytes": int((H + 1) * 24),
                "note": "SURVEY 8(d) tick latency: entry of pmaf_tick -> best index, set-point AND the selected agent's "
                        "scored path on the host (mapped pinned memory written by the manager kernel, no copy of the "
                        "other agents' paths); library clock, tick issued on an idle stream"},
            "allgather_us": head["allgather_us"],
            "header_exchange_us": head["header_exchange_us"],
            "roofline": {"bound": "hbm", "achieved": head["hbm_achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": head["hbm_frac"], "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name,
                         "avg_kernel_us": head["avg_kernel_us"],
                         "kernel_timing": head["kernel_timing"],
                         "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"],
                         "note": "FP64-VALU/latency-bound ODE integration; HBM fraction is structurally tiny "
                                 "(SURVEY.md 8d): see fp64_valu"},
            "fp64_valu": {"achieved_tflops": flops / avg_kernel_s / 1e12, "peak_tflops": FP64_VALU_PEAK_TF,
                          "frac": flops / avg_kernel_s / 1e12 / FP64_VALU_PEAK_TF,
                          "flops_per_agent_step": flops_step, "flops_source": flop_src,
                          "flops_per_agent_step_measured": fl.get("flops_per_agent_step"),
                          "in_shell_step_fraction": fl.get("in_shell_step_fraction"),
                          "flop_count": fl,
                          "pmc": pmc, "pmc_source": traffic_src},
        }
        if subs:
            out["configs"] = subs
        c5 = subs.get("C5_sharded")
        if c5 and "rollouts_per_s" in c5:
            # BASELINE config 5's STRONG-scaling record, lifted to the top level beside the weak-scaling `value` (C2, one
            # population per GPU, ~N x by construction) so that a reader of the 1/2/4/8-GPU lines finds the scaling
            # statement SURVEY 8(e) asks for without digging through `configs`
            sc5 = {"read_this_for_scaling": "`value` is BASELINE C2 replicated per GPU (weak scaling: the same 64-agent scene on every "
                                            "GPU). The scaling record of BASELINE config 5 (8 scenes x 1024 agents, strong scaling) is THIS block.",
                   "workload": c5["workload"], "scaling": "strong", "n_gpus": world,
                   "measured": {"rollouts_per_s": c5["rollouts_per_s"], "ms_per_tick": c5["ms_per_tick"],
                                "avg_kernel_us": c5["avg_kernel_us"], "populations_per_gpu": c5["populations_per_gpu"],
                                "lanes_per_agent": c5["lanes_per_agent"], "allgather_us": c5.get("allgather_us"),
                                # SURVEY 8(e)'s scaling report in one place: aggregate rollouts/s and ms per tick above, the
                                # all-gather beside them, per-GPU tick latency and kernel time here, efficiency below
                                "per_gpu_tick_us": c5["tick_latency_us"]["per_rank_median"],
                                "per_gpu_kernel_us": c5["per_rank_kernel_us"]},
                   "reason": C5_SCALING_REASON}
            one = subs.get("C5_one_gpu")
            if one and "rollouts_per_s" in one:
                sc5["one_gpu_same_job"] = {"rollouts_per_s": one["rollouts_per_s"], "ms_per_tick": one["ms_per_tick"],
                                           "avg_kernel_us": one["avg_kernel_us"], "lanes_per_agent": one["lanes_per_agent"]}
                sc5["measured"]["speedup_vs_one_gpu_same_job"] = c5["rollouts_per_s"] / one["rollouts_per_s"]
                sc5["measured"]["efficiency_vs_one_gpu_same_job"] = c5["rollouts_per_s"] / one["rollouts_per_s"] / world
            if emu is not None:
                sc5["prediction"] = emu
                c5["predicted_by_n"] = emu["predicted_by_n"]
            else:
                sc5["prediction"] = committed_c5_prediction()
            if sc5.get("prediction") and str(world) in sc5["prediction"]["predicted_by_n"]:
                pr = sc5["prediction"]["predicted_by_n"][str(world)]
                sc5["measured"]["vs_predicted_ms_per_tick"] = c5["ms_per_tick"] / pr["ms_per_tick"]
            out["scaling_c5"] = sc5
        if world == 1 and not args.only_headline:
            # tick latency where the reference's node would see it: through the C++ facade, open / closed loop / five calls
            out["cxx_boundary_latency_us"] = cxx_boundary_latency()
        if args.cpu_seconds > 0 and world == 1:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.config if args.config in ("C1", "C2", "C3", "C5") else "C2", args.cpu_seconds)
            ts = subs.get("task_static1")
            if ts and "rollouts_per_s" in ts:   # the CPU port at the reference's own parallelism: one thread per agent
                cb = cpu_baseline("task_static1", min(args.cpu_seconds, 8.0))
                if "error" not in cb:
                    ts["cpu_port"] = {k: cb[k] for k in ("value", "best", "cores", "unit", "h_eff", "value_1core", "best_1core",
                                                         "cpu_model", "share_of_repetitions_within_10pct_of_best")}
                    ts["cpu_port"]["tick_us_best"] = 1e6 * ts["agents"] / cb["best"]
                    ts["cpu_port"]["gpu_over_cpu_best"] = ts["rollouts_per_s"] / cb["best"]
                else:
                    ts["cpu_port"] = cb
        else:
            out["cpu_baseline"] = None
        line = json.dumps(out)

    def flush_all():
        sys.stdout.flush()
        try:  # ... including what C libraries (RCCL's banner) still hold in their stdio buffers
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass

    for comm, _ in ctx.comms.values():
        if comm is not None:
            comm.close()
    # the JSON line is the last thing written by the job: every rank empties its
    # buffers before the final barrier, rank 0 prints after it
    flush_all()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        flush_all()
        print(line, flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- planner-tick throughput of the MI355X-native circular-field
planner (BASELINE.json metric: agent-rollouts/s + planner tick latency).

One "step" = one planner tick (pmaf_tick: stop -> evaluateAgents ->
moveRealEEAgent -> resetEEAgents -> startPrediction) of every population of
the workload. Obstacles / agent state are resident in HBM before the timed
region; ticks are issued back to back, each returning best index + next
set-point to the host (the real per-tick API, not a batched open-loop shortcut).

Workloads
  default          BASELINE C2 (the metric's config): 64 agents x 200 steps x 32
                   obstacles, one population per GPU. N > 1 (one rank per GPU,
                   launched by torch.distributed.run): every rank plans its own
                   population ("weak" scaling) AND the ranks all-gather their
                   winner records once per tick -- ncclAllGather (RCCL over
                   xGMI) enqueued by libpmaf_hip.so on a second stream, so the
                   sharded run's collective is part of the measurement.
  --config C5 --shard   BASELINE C5: 8 goal/obstacle scenes x 1024 agents,
                   scenes {s : s % N == r} on rank r ("strong" scaling: the 8
                   scenes are fixed), winner all-gather per tick.
  --config C4      BASELINE C4: dual arm, 2 x 256 agents, each arm's repulsive
                   sphere follows the other arm's set-point. 2 ranks: one arm per
                   GPU, the set-points travel in the winner records; 1 rank: both
                   arms in one handle.
  --config C1|C3, --populations P, --dynamic   other single-GPU shapes.

Timing: W warm-up ticks, then blocks of K ticks, each block bracketed by a
barrier + device synchronisation on both sides; blocks are repeated until
--min-seconds have been timed (and at least --min-blocks blocks) and the
MEDIAN block (max over ranks) is reported, so a short `--steps 20` run gives
the stationary figure too.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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


def cpu_baseline(pkg, scene, budget_s, max_threads):
    """Times the CPU oracle (oracle/, a scalar C restatement of the
    reference's algorithm = kind 'port', gcc -O2) on this host: the same tick
    sequence on the same scene -- once on one core, and with the agents'
    rollouts spread over OpenMP threads (the reference's own parallelism is one
    thread per agent). Thread counts 8..max are tried for a bounded time each
    and the best one is reported with its count as `cores` (more threads than
    ~16 lose to fork/join cost on this 0.1 ms-per-rollout workload)."""
    os.environ.setdefault("OMP_WAIT_POLICY", "active")
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle import orc
    orc.set_exp_mode(0)
    N = scene["n_agents"]
    obs, dt, cg, ws = scene["obstacles"], scene["dt"], scene["cost_gains"], scene["ws_limits"]

    def run(nthreads, budget):
        o = orc.OraclePlanner(scene, mgr_init_pos=scene["start"])
        ticks, el = 0, 0.0
        t0 = time.perf_counter()
        while el < budget:
            if ticks % 256 == 0:
                o.set_initial_position(scene["start"])  # same episodes as the GPU run
            for _ in range(8):
                o.tick_omp(obs, dt, cg, ws, nthreads)
            ticks += 8
            el = time.perf_counter() - t0
        o.close()
        return N * ticks / el, ticks, el

    v1, _, _ = run(1, budget_s * 0.3)
    cands = [t for t in (8, 16, 32, 64, 128) if t <= max_threads] or [max_threads]
    best = None
    for t in cands:
        v, ticks, el = run(t, budget_s * 0.7 / len(cands))
        if best is None or v > best[0]:
            best = (v, t, ticks, el)
    return {
        "value": best[0], "unit": "rollouts/s", "cores": best[1], "kind": "port",
        "sample": "%d ticks (%.1f s) of the bench workload through oracle/libpmaf_oracle.so (gcc -O2 scalar C "
                  "restatement; rollouts of the %d agents on %d OpenMP threads, best of %s threads, "
                  "OMP_PROC_BIND=close; rest of the tick serial)" % (best[2], best[3], N, best[1], cands),
        "value_1core": v1, "host_cpus": os.cpu_count(),
    }


def kernel_name_of(cfg, n_obs):
    tiles = (n_obs - 1 + 63) // 64
    if 62 <= n_obs - 1 <= 64:
        tiles = 2
    generic = os.environ.get("PMAF_FORCE_GENERIC") == "1"
    if cfg["lanes_per_agent"] == 64 and tiles <= 4 and not generic:
        t = 1 if tiles <= 1 else 2 if tiles == 2 else 4
        dpp = t > 1 or (n_obs - 1) > 20
        if os.environ.get("PMAF_SUM"):
            dpp = t > 1 or os.environ["PMAF_SUM"].startswith("d")
        return "k_rollout_w64<%d, 2, %s>" % (t, "true" if dpp else "false")  # <TILES, MATH_XACT, DPPSUM>
    if cfg["lanes_per_agent"] in (8, 16, 32) and (n_obs - 2) // cfg["lanes_per_agent"] + 1 <= 4 and not generic:
        tl = (n_obs - 2) // cfg["lanes_per_agent"] + 1
        return "k_rollout_grp<%d, %d, 2>" % (cfg["lanes_per_agent"], 1 if tl <= 1 else 2 if tl == 2 else 4)
    return "k_rollout<%d>" % cfg["lanes_per_agent"]


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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget (0 = skip)")
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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # test hooks for a 1-GPU box: PMAF_BENCH_BACKEND=gloo runs torch's collectives on CPU tensors and the winner
    # exchange over a host-transport communicator, PMAF_BENCH_SINGLE_DEVICE=1 maps every rank to device 0
    backend = os.environ.get("PMAF_BENCH_BACKEND", "nccl")
    if os.environ.get("PMAF_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    red_dev = "cuda" if backend == "nccl" else "cpu"
    # PMAF_BENCH_FORCE_DIST=1: torch.distributed + an RCCL communicator even for one rank
    # -- exercises RCCL and this library's HIP runtime in one process on a 1-GPU box
    force_dist = os.environ.get("PMAF_BENCH_FORCE_DIST") == "1"
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    pkg = graft.load_package()
    pkg.load_library()
    S = pkg.scenes

    # ---- workload ----
    coupling = None
    scaling = "weak"
    if args.config == "C4":
        arms = S.dual_arm_scenes()
        if world == 2:
            mine = [rank]
        elif world == 1:
            mine = [0, 1]
        else:
            raise SystemExit("--config C4 runs on 1 GPU (both arms in one handle) or 2 GPUs (one arm per GPU)")
        scenes = [arms[a] for a in mine]
        total_pops = 2
        scaling = "strong"
        coupling = pkg.shard.DualArmCoupling(np.stack([s["obstacles"] for s in arms]), 0.1)
        workload = "C4 dual arm"
    elif args.shard:
        total_pops = args.total_populations
        if total_pops % world:
            raise SystemExit("--shard: --total-populations must be a multiple of the number of GPUs")
        mine = pkg.shard.partition_populations(total_pops, world, rank)
        scenes = [S.config_scene(args.config, scene_id=s, dynamic=args.dynamic) for s in mine]
        scaling = "strong"
        workload = "%s sharded" % args.config
    else:
        # weak scaling = the SAME work on every GPU: all ranks plan the same scene(s) unless --distinct-scenes
        # (the seeded scenes differ by +-3 % in tick time, which would read as a scaling loss of the slowest one)
        first = rank * args.populations if args.distinct_scenes else 0
        mine = list(range(first, first + args.populations))
        scenes = [S.config_scene(args.config, scene_id=s, dynamic=args.dynamic) for s in mine]
        total_pops = args.populations * world
        workload = args.config
    sc = scenes[0]
    N, H, n_obs = sc["n_agents"], sc["max_prediction_steps"] - 1, sc["obstacles"].shape[0]
    P = len(scenes)
    starts = np.stack([s["start"] for s in scenes])
    planner = pkg.PmafPlanner(scenes, device=local_rank, lanes_per_agent=args.lanes_per_agent, mgr_init_pos=starts)
    planner.set_initial_position(starts)
    obs = np.stack([s["obstacles"] for s in scenes])
    dt, cg, ws = sc["dt"], sc["cost_gains"], sc["ws_limits"]

    # ---- the sharded runs' collective: winner records all-gathered once per tick ----
    comm = None
    exchange = (world > 1 or force_dist) and not args.no_exchange
    transport = None
    if exchange:
        transport = "rccl" if backend == "nccl" else "host"
        if transport == "rccl":
            # the library's own RCCL communicator; should its bootstrap fail on any rank (environment), every rank
            # falls back to the host transport over a gloo side group so that the run still measures the exchange
            import torch
            ok = 1
            try:
                if os.environ.get("PMAF_BENCH_FAIL_RCCL") == "1":   # test hook for the fallback below
                    raise RuntimeError("PMAF_BENCH_FAIL_RCCL")
                comm = pkg.shard.make_comm(dist, world, rank, backend="rccl", device=local_rank)
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("rank %d: RCCL communicator failed (%s)\n" % (rank, e))
                comm, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int32, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.close()
                transport = "host"
                side = dist.new_group(backend="gloo")

                class _Side:  # the few calls torch_host_allgather makes, bound to the gloo group
                    @staticmethod
                    def get_world_size():
                        return world

                    @staticmethod
                    def all_gather_into_tensor(out, t):
                        return dist.all_gather_into_tensor(out, t, group=side)
                comm = pkg.PmafComm.host(world, rank, pkg.shard.torch_host_allgather(_Side))
        else:
            comm = pkg.shard.make_comm(dist, world, rank, backend="host", device=local_rank)
        planner.attach_comm(comm)

    tick_no = [0]
    arm_pos = [np.stack([a["start"] for a in S.dual_arm_scenes()])] if coupling is not None else None

    def one_tick(o):
        # stationary workload: restart the episode before the real agent gets
        # so close to the goal that rollouts stop early (cf_agent.cpp:310)
        if args.episode and tick_no[0] % args.episode == 0:
            planner.set_initial_position(starts)
            if coupling is not None:
                arm_pos[0] = np.stack([a["start"] for a in S.dual_arm_scenes()])
        tick_no[0] += 1
        if coupling is not None:
            # each arm's repulsive sphere = the other arm's last set-point; with one arm per GPU the set-points
            # come out of the winner records all-gathered behind the previous tick
            o = coupling.coupled_obstacles(arm_pos[0])[mine]
            b = planner.tick(o, dt, cg, ws)
            if comm is not None:
                tab = planner.winners_wait()          # [world][P][rec]
                arm_pos[0] = tab[:, 0, 4:7].copy()
            else:
                arm_pos[0] = planner.last_next_pos.copy()
            return b
        return planner.tick(o if args.dynamic else None, dt, cg, ws)

    def sync_all():
        planner.stop()
        if comm is not None:
            planner.winners_wait()
        if dist is not None:
            import torch
            if red_dev == "cuda":
                torch.cuda.synchronize()
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    obs0 = obs.copy()
    planner.tick(obs, dt, cg, ws)  # obstacles resident in HBM from here on
    for _ in range(args.warmup):
        one_tick(obs)
    planner.set_profiling(True)
    sync_all()
    planner.reset_kernel_stats()
    planner.exchange_times_us()  # clear

    # ---- timed region: blocks of `steps` ticks, barrier + device sync on both sides of each ----
    block_s, lat = [], []
    total_timed, n_blocks, max_blocks = 0.0, 0, 10000
    while True:
        blk_lat = np.zeros(args.steps)
        t0 = time.perf_counter()
        for k in range(args.steps):
            ta = time.perf_counter()
            one_tick(obs)
            blk_lat[k] = time.perf_counter() - ta
            if args.dynamic:
                # moving obstacles: advanced like dynamic_obstacle_node does, put back with the agent at every
                # episode start so the workload stays stationary (they would drift out of the scene otherwise)
                if args.episode and tick_no[0] % args.episode == 0:
                    obs = obs0.copy()
                else:
                    obs = np.stack([S.advance_live_obstacles(o) for o in obs])
        planner.stop()
        if comm is not None:
            planner.winners_wait()
        if dist is not None and red_dev == "cuda":
            import torch
            torch.cuda.synchronize()
        el = max_over_ranks(time.perf_counter() - t0)   # the all-reduce is the closing barrier of the block
        block_s.append(el)
        lat.append(blk_lat)
        total_timed += el
        n_blocks += 1
        # same decision on every rank (el is reduced). At least --min-blocks blocks, so that the median is one of
        # several blocks and a single slow block (a clock or scheduling hiccup on the box) cannot move it
        if (total_timed >= args.min_seconds and n_blocks >= args.min_blocks) or n_blocks >= max_blocks:
            break
        sync_all()
    elapsed = float(np.median(block_s))
    lat = np.concatenate(lat)
    kernel_ms, launches, agent_steps = planner.kernel_stats()
    cfg = planner.launch_config()
    ag_us = planner.exchange_times_us() if comm is not None else np.zeros(0)
    # set-point latency of a tick issued on an idle stream (the previous rollout
    # has finished, as in a 100 Hz control loop): host call -> best index and
    # next set-point on the host. Outside the timed region.
    idle = np.zeros(100)
    for k in range(idle.size):
        planner.stop()
        ta = time.perf_counter()
        one_tick(obs)
        idle[k] = time.perf_counter() - ta
    planner.stop()
    per_rank_tick_us = [float(np.median(lat) * 1e6)]
    per_rank_ag_us = [float(np.median(ag_us)) if ag_us.size else None]
    if dist is not None and world > 1:
        box = [None] * world
        dist.all_gather_object(box, (per_rank_tick_us[0], per_rank_ag_us[0]))
        per_rank_tick_us = [b[0] for b in box]
        per_rank_ag_us = [b[1] for b in box]
    if comm is not None:
        planner.attach_comm(None)
        comm.close()
    planner.close()

    if rank == 0:
        rollouts = N * total_pops * args.steps
        value = rollouts / elapsed
        avg_kernel_s = kernel_ms / max(launches, 1) * 1e-3
        bytes_per_launch = P * algorithmic_bytes_per_tick(N, H, n_obs)
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        steps_per_launch = agent_steps / max(launches, 1)
        kernel_name = kernel_name_of(cfg, n_obs)
        # exact flop count of this workload from the instrumented oracle
        fl = measured_flops(pkg, sc, args.flop_ticks) if args.flop_ticks > 0 else {"error": "skipped"}
        if "flops_per_agent_step" in fl:
            flops_step, flop_src = fl["flops_per_agent_step"], "measured"
        else:
            M = n_obs - 1   # SURVEY 8(d) estimate as the fall-back
            flops_step, flop_src = 110 + 47 * M + 12 * M + 65 * (M / 4.0), "SURVEY 8(d) estimate (flop count unavailable)"
        flops = flops_step * steps_per_launch
        # HBM traffic per launch of this kernel from the committed PMC passes
        # (tools/gpu_prof.sh + tools/traffic_from_pmc.py); PMC counters cannot be
        # collected from inside the timed run itself
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
            "metric": "agent_rollouts_per_s", "value": value, "unit": "rollouts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d agents x %d-step horizon, %d sphere obstacles + repulsive sentinel, "
                                   "%d population(s) per GPU (%d in the job), %s obstacles, one pmaf_tick per step%s"
                                   % (workload, N, H, n_obs - 1, P, total_pops, "moving" if args.dynamic else "static",
                                      ", winner records all-gathered once per tick (%s)" %
                                      ("RCCL" if transport == "rccl" else "host transport") if comm is not None else ""),
                       "agents": N, "horizon": H, "obstacles": n_obs - 1, "populations_per_gpu": P,
                       "populations_total": total_pops,
                       "parallelism": "population-per-gpu x%d (%s)" % (world, "sharded scenes" if scaling == "strong" else
                                                                      "distinct scenes" if args.distinct_scenes else
                                                                      "same scene on every GPU"),
                       "collective": None if comm is None else
                       "%s of %d B winner records per rank per tick, second stream, overlapped with the "
                       "rollout" % ("ncclAllGather" if transport == "rccl" else "host-transport all-gather",
                                    P * (8 + 3 * (H + 1)) * 8),
                       "lanes_per_agent": cfg["lanes_per_agent"], "rollout_blocks": cfg["n_blocks"],
                       "arithmetic": "f64, hand-expanded IEEE div/sqrt sequences (default policy; bit-identical to "
                                     "the CPU oracle)"},
            "timing": {"blocks": n_blocks, "block_ticks": args.steps, "timed_s": total_timed,
                       "block_ms": {"median": elapsed * 1e3, "min": float(np.min(block_s)) * 1e3,
                                    "max": float(np.max(block_s)) * 1e3},
                       "note": "value / ms_per_step = MEDIAN block of `steps` ticks (max over ranks per block)"},
            "agent_steps_per_s": steps_per_launch * world * args.steps / elapsed,
            "h_eff": steps_per_launch / (N * P),
            "tick_latency_us": {"median": float(np.median(lat) * 1e6), "p99": float(np.percentile(lat, 99) * 1e6),
                                "per_rank_median": per_rank_tick_us,
                                "note": "back-to-back ticks: each call waits for the previous rollout"},
            "setpoint_latency_us": {"median": float(np.median(idle) * 1e6), "p99": float(np.percentile(idle, 99) * 1e6),
                                    "note": "tick issued on an idle stream: host call -> best index + next set-point "
                                            "on the host (the new rollout then runs asynchronously)"},
            "allgather_us": None if comm is None else {
                "median": float(np.median(ag_us)) if ag_us.size else None,
                "p99": float(np.percentile(ag_us, 99)) if ag_us.size else None,
                "n": int(ag_us.size), "per_rank_median": per_rank_ag_us,
                "transport": transport,
                "note": ("device time between the events around ncclAllGather on the exchange stream (includes the "
                         "wait for the slowest rank); off the rollout's critical path") if transport == "rccl" else
                        "host transport: wall time of the all-gather callback (gloo), run when the table is asked for"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name,
                         "avg_kernel_us": avg_kernel_s * 1e6,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
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
        if args.cpu_seconds > 0 and world == 1:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(pkg, sc, args.cpu_seconds, max(1, min(N, os.cpu_count() or 1)))
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

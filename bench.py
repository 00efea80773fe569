#!/usr/bin/env python3
"""bench.py -- planner-tick throughput of the MI355X-native circular-field
planner (BASELINE.json metric: agent-rollouts/s + planner tick latency).

One "step" = one planner tick (pmaf_tick: stop -> evaluateAgents ->
moveRealEEAgent -> resetEEAgents -> startPrediction) of one population in the
BASELINE configuration the metric is quoted on:
    C2 = 64 agents, 200-step horizon, 32 synthetic sphere obstacles (+ the
    trailing repulsive obstacle), SURVEY.md 8(d) scene generator.
Obstacles / agent state are resident in HBM before the timed region; ticks are
issued back to back, each returning best index + next set-point to the host
(the real per-tick API, not a batched open-loop shortcut).

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank plans
its own independent population (the same scene on every GPU, so the per-GPU
work is fixed; --distinct-scenes gives rank r scene r) -- the path shards by
population with no data-path collective (weak scaling). torch.distributed is
used only for the barrier and the max-over-ranks timing.

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


def algorithmic_flops_per_agent_step(M):
    """SURVEY.md 8(d) estimate: 110 + 47 M (+12 M scaling sweep) + 65 S, with
    S (in-shell obstacles per step) taken as M/4."""
    return 110 + 47 * M + 12 * M + 65 * (M / 4.0)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="C2", help="C1|C2|C3|C5 (BASELINE.json configs); C2 is the metric's config")
    ap.add_argument("--populations", type=int, default=1, help="independent populations per GPU in one handle")
    ap.add_argument("--lanes-per-agent", type=int, default=0)
    ap.add_argument("--dynamic", action="store_true", help="moving obstacles, re-uploaded every tick")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget (0 = skip)")
    ap.add_argument("--distinct-scenes", action="store_true",
                    help="N > 1: rank r plans scene(s) r*P .. r*P+P-1 (a goal/obstacle sweep) instead of every rank "
                         "planning the same scene(s)")
    ap.add_argument("--episode", type=int, default=256,
                    help="ticks per episode: the real agent is put back at the start every EPISODE ticks so every "
                         "timed rollout runs its full horizon (stationary workload; 0 = never)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # test hooks for a 1-GPU box: PMAF_BENCH_BACKEND=gloo runs the collectives on
    # CPU tensors, PMAF_BENCH_SINGLE_DEVICE=1 maps every rank to device 0
    backend = os.environ.get("PMAF_BENCH_BACKEND", "nccl")
    if os.environ.get("PMAF_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    red_dev = "cuda" if backend == "nccl" else "cpu"
    # PMAF_BENCH_FORCE_DIST=1: initialise torch.distributed (RCCL) even for one rank
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
    # weak scaling = the SAME work on every GPU: all ranks plan the same scene(s) unless --distinct-scenes
    # (the seeded scenes differ by +-3 % in tick time, which would read as a scaling loss of the slowest one)
    first = rank * args.populations if args.distinct_scenes else 0
    scenes = [pkg.scenes.config_scene(args.config, scene_id=first + i, dynamic=args.dynamic)
              for i in range(args.populations)]
    sc = scenes[0]
    N, H, n_obs = sc["n_agents"], sc["max_prediction_steps"] - 1, sc["obstacles"].shape[0]
    P = args.populations
    starts = np.stack([s["start"] for s in scenes])
    planner = pkg.PmafPlanner(scenes, device=local_rank, lanes_per_agent=args.lanes_per_agent, mgr_init_pos=starts)
    planner.set_initial_position(starts)
    obs = np.stack([s["obstacles"] for s in scenes])
    dt, cg, ws = sc["dt"], sc["cost_gains"], sc["ws_limits"]

    tick_no = [0]

    def one_tick(o):
        # stationary workload: restart the episode before the real agent gets
        # so close to the goal that rollouts stop early (cf_agent.cpp:310)
        if args.episode and tick_no[0] % args.episode == 0:
            planner.set_initial_position(starts)
        tick_no[0] += 1
        return planner.tick(o if args.dynamic else None, dt, cg, ws)

    def sync_all():
        planner.stop()
        if dist is not None:
            import torch
            if red_dev == "cuda":
                torch.cuda.synchronize()
            dist.barrier()

    obs0 = obs.copy()
    planner.tick(obs, dt, cg, ws)  # obstacles resident in HBM from here on
    for _ in range(args.warmup):
        one_tick(obs)
    planner.set_profiling(True)
    sync_all()
    planner.reset_kernel_stats()
    lat = np.zeros(args.steps)
    t0 = time.perf_counter()
    for k in range(args.steps):
        ta = time.perf_counter()
        one_tick(obs)
        lat[k] = time.perf_counter() - ta
        if args.dynamic:
            # moving obstacles: advanced like dynamic_obstacle_node does, put back with the agent at every
            # episode start so the workload stays stationary (they would drift out of the scene otherwise)
            if args.episode and tick_no[0] % args.episode == 0:
                obs = obs0.copy()
            else:
                obs = np.stack([pkg.scenes.advance_live_obstacles(o) for o in obs])
    planner.stop()
    if dist is not None and red_dev == "cuda":
        import torch
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        elapsed = float(t.item())
    kernel_ms, launches, agent_steps = planner.kernel_stats()
    cfg = planner.launch_config()
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
    planner.close()

    if rank == 0:
        rollouts = N * P * world * args.steps
        value = rollouts / elapsed
        avg_kernel_s = kernel_ms / max(launches, 1) * 1e-3
        bytes_per_launch = P * algorithmic_bytes_per_tick(N, H, n_obs)
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        steps_per_launch = agent_steps / max(launches, 1)
        flops = algorithmic_flops_per_agent_step(n_obs - 1) * steps_per_launch
        tiles = (n_obs - 1 + 63) // 64
        if cfg["lanes_per_agent"] == 64 and tiles <= 4 and os.environ.get("PMAF_FORCE_GENERIC") != "1":
            kernel_name = "k_rollout_w64<%d, 2>" % (1 if tiles <= 1 else 2 if tiles == 2 else 4)  # <TILES, MATH_XACT>
        elif cfg["lanes_per_agent"] in (8, 16, 32) and (n_obs - 2) // cfg["lanes_per_agent"] + 1 <= 4 \
                and os.environ.get("PMAF_FORCE_GENERIC") != "1":
            tl = (n_obs - 2) // cfg["lanes_per_agent"] + 1
            kernel_name = "k_rollout_grp<%d, %d, 2>" % (cfg["lanes_per_agent"], 1 if tl <= 1 else 2 if tl == 2 else 4)
        else:
            kernel_name = "k_rollout<%d>" % cfg["lanes_per_agent"]
        # HBM traffic per launch of this kernel from the committed PMC passes
        # (tools/gpu_prof.sh + tools/traffic_from_pmc.py); PMC counters cannot be
        # collected from inside the timed run itself
        traffic = None
        pmc = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            key = "%s%s:%s" % (args.config, "" if P == 1 else "x%d" % P, kernel_name)
            if key in tj and not args.dynamic:
                traffic = tj[key]["traffic_bytes_per_launch"]
                pmc = {k: tj[key].get(k) for k in ("valu_active_frac_of_wave_cycles", "valu_insts_per_launch", "waves_per_launch")}
        except (OSError, ValueError, KeyError):
            traffic = None
        out = {
            "metric": "agent_rollouts_per_s", "value": value, "unit": "rollouts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d agents x %d-step horizon, %d sphere obstacles + repulsive sentinel, "
                                   "%d population(s) per GPU, %s obstacles, one pmaf_tick per step"
                                   % (args.config, N, H, n_obs - 1, P, "moving" if args.dynamic else "static"),
                       "agents": N, "horizon": H, "obstacles": n_obs - 1, "populations_per_gpu": P,
                       "parallelism": "population-per-gpu x%d (%s)" % (world, "distinct scenes" if args.distinct_scenes
                                                                      else "same scene on every GPU"),
                       "lanes_per_agent": cfg["lanes_per_agent"], "rollout_blocks": cfg["n_blocks"],
                       "arithmetic": "f64, hand-expanded IEEE div/sqrt sequences (default policy; bit-identical to "
                                     "the CPU oracle)"},
            "agent_steps_per_s": agent_steps / max(launches, 1) * world * args.steps / elapsed,
            "h_eff": steps_per_launch / (N * P),
            "tick_latency_us": {"median": float(np.median(lat) * 1e6), "p99": float(np.percentile(lat, 99) * 1e6),
                                "note": "back-to-back ticks: each call waits for the previous rollout"},
            "setpoint_latency_us": {"median": float(np.median(idle) * 1e6), "p99": float(np.percentile(idle, 99) * 1e6),
                                    "note": "tick issued on an idle stream: host call -> best index + next set-point "
                                            "on the host (the new rollout then runs asynchronously)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name,
                         "avg_kernel_us": avg_kernel_s * 1e6,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "FP64-VALU/latency-bound ODE integration; HBM fraction is structurally tiny "
                                 "(SURVEY.md 8d): see fp64_valu"},
            "fp64_valu": {"achieved_tflops": flops / avg_kernel_s / 1e12, "peak_tflops": FP64_VALU_PEAK_TF,
                          "frac": flops / avg_kernel_s / 1e12 / FP64_VALU_PEAK_TF,
                          "flops_per_agent_step_est": algorithmic_flops_per_agent_step(n_obs - 1),
                          "pmc": pmc},  # from the committed PMC passes (profiles/traffic.json), like roofline.traffic
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

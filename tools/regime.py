"""Where the MI355X planner wins and where it loses: per-step chain latency of one rollout on the
GPU against one x86 core, and the agent count at which one MI355X overtakes a host that runs one agent per core, for
M = 9 / 32 / 128 field obstacles.

GPU: rollout-kernel time per launch (HIP events on the kernel's own dispatch) of N agents x H steps through M synthetic
obstacles (SURVEY 8d scenes, far goal: every rollout runs its full horizon), for N = 8 ... 8192. A launch lasts as long
as its longest chain while every wave has a SIMD of its own, so kernel_us / H is the chain latency per step.
CPU: the oracle (TEST INFRASTRUCTURE, used here as the stated baseline only) on ONE pinned core: seconds per
agent-step; a host with C cores running one agent per core needs ceil(N / C) x H x that per tick.
Writes a JSON record and prints the table DESIGN.md quotes.
usage: python tools/regime.py [--out profiles/r6_regime.json] [--horizon 200]"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--horizon", type=int, default=200)
ap.add_argument("--cpu-seconds", type=float, default=3.0)
args = ap.parse_args()
pm = g.load_package()
from oracle import orc  # noqa: E402  (the stated CPU baseline)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cpu_bench  # noqa: E402

cores, n_aff, quota = cpu_bench.allowed_cpus()
aff0 = os.sched_getaffinity(0)
orc.set_exp_mode(0)
H = args.horizon
rec = {"horizon": H, "cpu_model": cpu_bench.cpu_model(), "physical_cores_allowed": len(cores), "rows": []}
for M in (9, 32, 128):
    # ---- one CPU core: seconds per agent-step
    os.sched_setaffinity(0, {cores[0]})      # the one-core measurement on one pinned core
    sc = pm.scenes.synthetic_scene(16, H, M, 2, 0)
    o = orc.OraclePlanner(sc, mgr_init_pos=sc["start"])
    o.set_initial_position(sc["start"])
    for _ in range(3):
        o.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
    best, t_end, s0 = 1e9, time.perf_counter() + args.cpu_seconds, o.agent_steps()
    while time.perf_counter() < t_end:
        a0, t0 = o.agent_steps(), time.perf_counter()
        for _ in range(4):
            o.tick(sc["obstacles"], sc["dt"], sc["cost_gains"], sc["ws_limits"])
        dt_ = time.perf_counter() - t0
        best = min(best, dt_ / (o.agent_steps() - a0))
        if o.agent_steps() - s0 > 16 * H * 40:      # put the real agent back: full-horizon rollouts only
            o.set_initial_position(sc["start"])
            s0 = o.agent_steps()
    o.close()
    os.sched_setaffinity(0, aff0)
    cpu_ns = best * 1e9
    # ---- GPU: kernel time per launch over the agent count
    gpu = {}
    for N in (8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
        scn = pm.scenes.synthetic_scene(N, H, M, 2, 0)
        h = pm.PmafPlanner(scn, device=0, mgr_init_pos=scn["start"])
        h.set_initial_position(scn["start"])
        h.set_profiling(True)
        for _ in range(8):
            h.tick(None, scn["dt"], scn["cost_gains"], scn["ws_limits"])
        h.stop()
        h.reset_kernel_stats()
        K = 40 if N <= 1024 else 16
        for k in range(K):
            if k % 32 == 0:
                h.set_initial_position(scn["start"])
            h.tick(None, scn["dt"], scn["cost_gains"], scn["ws_limits"])
        h.stop()
        ms, n, steps = h.kernel_stats()
        cfg = h.launch_config()
        gpu[N] = dict(kernel_us=ms / n * 1e3, h_eff=steps / n / N, lanes_per_agent=cfg["lanes_per_agent"], waves_per_agent=cfg.get("waves_per_agent", 1))
        h.close()

    def cpu_tick_us(N, C):
        return math.ceil(N / C) * H * cpu_ns * 1e-3

    cross = {}
    for C in sorted({64, len(cores)}):
        c = next((N for N in sorted(gpu) if gpu[N]["kernel_us"] < cpu_tick_us(N, C)), None)
        cross[str(C)] = c
    row = dict(field_obstacles=M, cpu_ns_per_agent_step_one_core=cpu_ns, gpu_ns_per_step_of_a_chain=gpu[64]["kernel_us"] * 1e3 / H,
               gpu_chain_over_cpu_chain=gpu[64]["kernel_us"] * 1e3 / H / cpu_ns, gpu_kernel_us_by_agents={str(k): v for k, v in gpu.items()},
               agents_at_which_the_gpu_overtakes_a_host_of_C_cores=cross,
               gpu_agent_steps_per_us_at_8192=8192 * H / gpu[8192]["kernel_us"], cpu_agent_steps_per_us_64_cores=64 / (cpu_ns * 1e-3))
    rec["rows"].append(row)
    print("M = %3d: one core %.0f ns / agent-step | GPU chain %.0f ns / step (%.1f x slower per chain) | GPU kernel us for N = 8..8192: %s | overtakes C cores at N = %s | "
          "aggregate at N = 8192: GPU %.0f vs 64 cores %.0f agent-steps/us" % (
              M, cpu_ns, row["gpu_ns_per_step_of_a_chain"], row["gpu_chain_over_cpu_chain"],
              " ".join("%.0f" % gpu[N]["kernel_us"] for N in sorted(gpu)), cross, row["gpu_agent_steps_per_us_at_8192"],
              row["cpu_agent_steps_per_us_64_cores"]), flush=True)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(rec, open(args.out, "w"), indent=1)

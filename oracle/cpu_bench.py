"""CPU baseline of bench.py: times the CPU oracle -- TEST
INFRASTRUCTURE, the scalar C restatement of the reference's algorithm, kind
"port" -- on this host, in a process of its own (one oracle build per process,
own OpenMP settings). Prints one JSON object.

Per build (`-O2`, and `-O3 -march=native` compiled on THIS machine): the bench's
own tick sequence on the bench's own scene,
  * on ONE core: >= `--reps` repetitions of a fixed number of ticks, median /
    min / max rollouts/s;
  * with the agents' rollouts on OpenMP threads (the reference's parallelism is
    one thread per agent, B/src/cf_manager.cpp:118-123; the instrument mirrored
    is CfAgent::prediction_time_, B/src/cf_agent.cpp:308-331): a short probe
    per thread count, then >= `--reps` repetitions at the count whose probe
    median was best -- median / min / max reported, not best-of.
usage: python oracle/cpu_bench.py --config C2 [--reps 30] [--budget 11]"""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def one_build(build, args):
    """runs in a child process: PMAF_ORACLE_LIB already points at the build"""
    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    from oracle import orc
    pkg = graft.load_package()
    orc.set_exp_mode(0)
    scene = pkg.scenes.config_scene(args.config)
    N = scene["n_agents"]
    obs, dt, cg, ws = scene["obstacles"], scene["dt"], scene["cost_gains"], scene["ws_limits"]

    def planner():
        o = orc.OraclePlanner(scene, mgr_init_pos=scene["start"])
        o.set_initial_position(scene["start"])
        return o

    def fingerprint():
        o = planner()
        b = [o.tick(obs, dt, cg, ws) for _ in range(3)]
        pos = o.real_state()[0].copy()
        paths, n = o.paths()
        o.close()
        return b, pos.tobytes().hex(), float(paths.sum()), int(n.sum())

    def reps(nthreads, n_reps, ticks, deadline):
        o = planner()
        for _ in range(2):
            o.tick_omp(obs, dt, cg, ws, nthreads)
        vals, done = [], 2
        for r in range(n_reps):
            if done + ticks > args.episode:       # same episodes as the GPU run: full-horizon rollouts only
                o.set_initial_position(scene["start"])
                done = 0
            t0 = time.perf_counter()
            for _ in range(ticks):
                o.tick_omp(obs, dt, cg, ws, nthreads)
            vals.append(N * ticks / (time.perf_counter() - t0))
            done += ticks
            if time.perf_counter() > deadline and len(vals) >= 5:
                break
        h_eff = o.agent_steps() / float(N * (2 + sum([ticks] * len(vals)))) if hasattr(o, "agent_steps") else None
        o.close()
        v = np.asarray(vals)
        return dict(h_eff=h_eff, median=float(np.median(v)), min=float(v.min()), max=float(v.max()), reps=int(v.size), ticks_per_rep=ticks)

    t_start = time.perf_counter()
    budget = args.budget
    # one core: size the repetition so that `reps` of them take ~35 % of the budget
    o = planner()
    t0 = time.perf_counter()
    o.tick_omp(obs, dt, cg, ws, 1)
    t_tick = time.perf_counter() - t0
    o.close()
    ticks1 = max(1, min(16, int(0.35 * budget / args.reps / t_tick)))
    one = reps(1, args.reps, ticks1, t_start + 0.45 * budget)
    cands = [t for t in (8, 16, 32, 64, 128) if t <= min(N, os.cpu_count() or 1)] or [min(N, os.cpu_count() or 1)]
    probe = {}
    for t in cands:
        probe[t] = reps(t, 6, max(4, ticks1 * 4), t_start + 0.45 * budget + 0.25 * budget * (cands.index(t) + 1) / len(cands))["median"]
    win = max(probe, key=probe.get)
    multi = reps(win, args.reps, max(4, ticks1 * 8), t_start + budget)
    return dict(build=build, one_core=one, threads=win, multi=multi, probe_median_by_threads={str(k): v for k, v in probe.items()},
                fingerprint=fingerprint(), seconds=time.perf_counter() - t_start)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--budget", type=float, default=11.0, help="seconds for both builds together")
    ap.add_argument("--episode", type=int, default=256, help="ticks after which the real agent is put back at the start")
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        args.budget = args.budget
        print(json.dumps(one_build(args.child, args)))
        return
    env = dict(os.environ)
    env.setdefault("OMP_WAIT_POLICY", "active")
    env.setdefault("OMP_PROC_BIND", "close")
    env.setdefault("OMP_PLACES", "cores")
    builds = {"O2": os.path.join(HERE, "libpmaf_oracle.so")}
    subprocess.check_call(["make", "-C", HERE, "-s", "libpmaf_oracle.so"])
    try:
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "_native/libpmaf_oracle_O3native.so"])
        builds["O3_native"] = os.path.join(HERE, "_native", "libpmaf_oracle_O3native.so")
    except (subprocess.CalledProcessError, OSError) as e:
        sys.stderr.write("cpu_bench: -O3 -march=native build failed: %s\n" % e)
    out = {"cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "builds": {}}
    for name, so in builds.items():
        e = dict(env, PMAF_ORACLE_LIB=so)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", args.config, "--reps", str(args.reps),
                            "--budget", str(args.budget / len(builds)), "--episode", str(args.episode), "--child", name],
                           capture_output=True, text=True, env=e, timeout=120 + 4 * args.budget)
        if r.returncode != 0:
            out["builds"][name] = {"error": r.stderr[-500:]}
            continue
        out["builds"][name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    fps = [json.dumps(b.get("fingerprint")) for b in out["builds"].values() if "fingerprint" in b]
    out["builds_bit_identical"] = len(set(fps)) == 1 and len(fps) == len(out["builds"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""CPU baseline of bench.py: times the CPU oracle -- TEST
INFRASTRUCTURE, the scalar C restatement of the reference's algorithm, kind
"port" -- on this host, in a process of its own (one oracle build per process,
own OpenMP settings). Prints one JSON object.

Made to REPRODUCE (the driver's records of rounds 3 / 4 show 254 k and 60 k rollouts/s for the same
workload, single repetitions from 5.6 k to 252 k):
  * the CPUs this process may use are read, not assumed: sched_getaffinity intersected with the cgroup's cpu.max quota,
    reduced to ONE logical CPU per physical core (thread_siblings_list) -- stated in the output;
  * the OpenMP threads are pinned explicitly to that list (GOMP_CPU_AFFINITY, OMP_PROC_BIND=true), one thread per core;
    never more threads than agents or allowed cores;
  * >= 3 s of multi-threaded warm-up per build (clocks, thread pool, page cache) before anything is timed;
  * the thread count is chosen by the BEST repetition of a probe per candidate, not by a probe median;
  * reported per build: best, median, min, max and the share of repetitions within 10 % of the best (1.0 = unimodal;
    on a shared host the distribution is bimodal -- a neighbour's burst halves a repetition -- and `best` is the
    number that reproduces from run to run). bench.py quotes `best` next to `value` (= median).

Per build (`-O2`, and `-O3 -march=native` compiled on THIS machine): the bench's own tick sequence on the bench's own
scene, on ONE pinned core and with the agents' rollouts on OpenMP threads (the reference's parallelism is one thread per
agent, B/src/cf_manager.cpp:118-123; the instrument mirrored is CfAgent::prediction_time_, B/src/cf_agent.cpp:308-331).
--config task_static1 = the reference's shipped operating point (dual_arms_static1.yaml: 10 agents,
max_prediction_steps 1500, 9 + 1 obstacles), one thread per agent as the reference runs it.
usage: python oracle/cpu_bench.py --config C2 [--reps 30] [--budget 20]"""
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


def allowed_cpus():
    """(one logical CPU per physical core out of this process's affinity set, the affinity set's size, the cgroup's
    CPU quota in cores or None)"""
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    quota = None
    try:   # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:   # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / float(p)
        except (OSError, ValueError):
            pass
    seen, cores = set(), []
    for c in aff:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            cores.append(c)
    if quota is not None and quota >= 1:
        cores = cores[:max(1, int(quota))]
    return cores, len(aff), quota


def scene_of(pkg, config):
    if config == "task_static1":
        return pkg.scenes.static1_scene(10, 1499)
    return pkg.scenes.config_scene(config)


def summary(vals, ticks):
    import numpy as np
    v = np.asarray(vals)
    best = float(v.max())
    return dict(best=best, median=float(np.median(v)), min=float(v.min()), max=best, reps=int(v.size), ticks_per_rep=ticks,
                share_within_10pct_of_best=float((v >= 0.9 * best).mean()))


def one_build(build, args):
    """runs in a child process: PMAF_ORACLE_LIB points at the build, GOMP_CPU_AFFINITY at the core list"""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    from oracle import orc
    pkg = graft.load_package()
    orc.set_exp_mode(0)
    scene = scene_of(pkg, args.config)
    N = scene["n_agents"]
    n_cores = args.cores
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

    def run(nthreads, ticks, deadline, max_reps, min_reps=5):
        o = planner()
        for _ in range(2):
            o.tick_omp(obs, dt, cg, ws, nthreads)
        vals, done, total_ticks = [], 2, 2
        while len(vals) < max_reps:
            if done + ticks > args.episode:       # same episodes as the GPU run
                o.set_initial_position(scene["start"])
                done = 0
            t0 = time.perf_counter()
            for _ in range(ticks):
                o.tick_omp(obs, dt, cg, ws, nthreads)
            vals.append(N * ticks / (time.perf_counter() - t0))
            done += ticks
            total_ticks += ticks
            if time.perf_counter() > deadline and len(vals) >= min_reps:
                break
        h_eff = o.agent_steps() / float(N * total_ticks)
        o.close()
        return vals, h_eff

    t_start = time.perf_counter()
    per = args.budget
    # candidates: never more threads than agents or allowed cores (one agent's rollout is one serial chain)
    top = max(1, min(N, n_cores))
    if args.config == "task_static1":
        cands = [top]                              # one thread per agent, as the reference runs
    else:
        cands = sorted({t for t in (2, 4, 8, 16, 32, 64, 128) if t <= top} | {top})
    # size one repetition from a single tick
    o = planner()
    t0 = time.perf_counter()
    o.tick_omp(obs, dt, cg, ws, 1)
    t_tick1 = time.perf_counter() - t0
    o.close()
    # ---- warm-up: >= 3 s of multi-threaded ticks at the widest candidate (thread pool up, clocks settled)
    warm_s = max(3.0, 0.15 * per)
    run(top, 4, time.perf_counter() + warm_s, 10 ** 9, min_reps=1)
    # ---- one pinned core
    ticks1 = max(1, min(16, int(0.20 * per / args.reps / t_tick1)))
    v1, _ = run(1, ticks1, time.perf_counter() + 0.20 * per, args.reps)
    one = summary(v1, ticks1)
    # ---- probe: the best repetition per candidate decides
    tm = max(4, int(ticks1 * min(top, 8)))
    probe = {}
    for t in cands:
        pv, _ = run(t, tm, time.perf_counter() + 0.20 * per / len(cands), 8, min_reps=4)
        probe[t] = max(pv)
    win = max(probe, key=probe.get)
    vm, h_eff = run(win, tm, t_start + per + warm_s, args.reps, min_reps=10)
    multi = summary(vm, tm)
    multi["h_eff"] = h_eff
    return dict(build=build, one_core=one, threads=win, multi=multi, probe_best_by_threads={str(k): v for k, v in probe.items()},
                warmup_s=warm_s, fingerprint=fingerprint(), seconds=time.perf_counter() - t_start)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--budget", type=float, default=20.0, help="seconds for both builds together (warm-up on top)")
    ap.add_argument("--episode", type=int, default=256, help="ticks after which the real agent is put back at the start")
    ap.add_argument("--child", default=None)
    ap.add_argument("--cores", type=int, default=0)
    args = ap.parse_args()
    if args.child:
        print(json.dumps(one_build(args.child, args)))
        return
    cores, n_aff, quota = allowed_cpus()
    env = dict(os.environ)
    env["OMP_WAIT_POLICY"] = "active"
    env["OMP_PROC_BIND"] = "true"
    env["GOMP_CPU_AFFINITY"] = " ".join(str(c) for c in cores)   # thread k on the k-th allowed physical core
    env.pop("OMP_PLACES", None)
    env["OMP_DYNAMIC"] = "false"
    builds = {"O2": os.path.join(HERE, "libpmaf_oracle.so")}
    subprocess.check_call(["make", "-C", HERE, "-s", "libpmaf_oracle.so"])
    try:
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "_native/libpmaf_oracle_O3native.so"])
        builds["O3_native"] = os.path.join(HERE, "_native", "libpmaf_oracle_O3native.so")
    except (subprocess.CalledProcessError, OSError) as e:
        sys.stderr.write("cpu_bench: -O3 -march=native build failed: %s\n" % e)
    out = {"cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "affinity_cpus": n_aff, "cgroup_cpu_quota": quota,
           "physical_cores_used": len(cores), "pinning": "GOMP_CPU_AFFINITY (one OpenMP thread per allowed physical core), "
           "OMP_PROC_BIND=true, OMP_WAIT_POLICY=active", "builds": {}}
    for name, so in builds.items():
        e = dict(env, PMAF_ORACLE_LIB=so, PMAF_VARIANT="")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", args.config, "--reps", str(args.reps),
                            "--budget", str(args.budget / len(builds)), "--episode", str(args.episode), "--child", name,
                            "--cores", str(len(cores))],
                           capture_output=True, text=True, env=e, timeout=180 + 4 * args.budget)
        if r.returncode != 0:
            out["builds"][name] = {"error": r.stderr[-500:]}
            continue
        out["builds"][name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    fps = [json.dumps(b.get("fingerprint")) for b in out["builds"].values() if "fingerprint" in b]
    out["builds_bit_identical"] = len(set(fps)) == 1 and len(fps) == len(out["builds"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()

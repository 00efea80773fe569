"""CPU tests of bench.py's multi-rank plumbing (no GPU, no planner): `--gpus N`
without a launcher starts N ranks itself, the ranks rendezvous, build the
exchange communicator and all-gather through it (--dry-run); a world that
differs from --gpus, or more ranks than devices, is an error -- never a silent
1-GPU line labelled otherwise."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if k not in env:
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=300, env=e, cwd=ROOT)


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dry-run"], PMAF_BENCH_BACKEND="gloo", PMAF_BENCH_SINGLE_DEVICE="1")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["collective_world"] == 2
    assert [w[0] for w in out["ranks"]] == [0, 1] and out["ranks"][0][2] != out["ranks"][1][2]   # two processes
    assert out["allgather_of_ranks"] == [0.0, 1.0]


def test_world_size_differing_from_gpus_is_refused():
    r = _run(["--gpus", "2", "--dry-run"], WORLD_SIZE="1", RANK="0", PMAF_BENCH_BACKEND="gloo", PMAF_BENCH_SINGLE_DEVICE="1")
    assert r.returncode != 0 and "refusing" in r.stderr
    r = _run(["--gpus", "1", "--dry-run"], WORLD_SIZE="2", RANK="0", PMAF_BENCH_BACKEND="gloo", PMAF_BENCH_SINGLE_DEVICE="1")
    assert r.returncode != 0 and "refusing" in r.stderr


def test_more_ranks_than_devices_is_refused_without_the_single_device_hook():
    # gloo backend so that the process group comes up without a GPU; no SINGLE_DEVICE hook: 2 ranks > devices visible
    r = _run(["--gpus", "2", "--dry-run"], PMAF_BENCH_BACKEND="gloo")
    assert r.returncode != 0 and "hipGetDeviceCount" in r.stderr


def test_gpus_8_dry_run_and_the_plan_budget():
    """the machine the driver scales on has 8 GPUs: `--gpus 8` without a launcher starts eight ranks (gloo, one device
    hook), the exchange communicator spans all eight, and the launch plan of the default command stays inside one
    minute of GPU-side time by construction of its sub-configuration budgets (bench.plan_budget_s)"""
    r = _run(["--gpus", "8", "--dry-run", "--steps", "20", "--warmup", "5"], PMAF_BENCH_BACKEND="gloo", PMAF_BENCH_SINGLE_DEVICE="1")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["collective_world"] == 8
    assert [w[0] for w in out["ranks"]] == list(range(8)) and len({w[2] for w in out["ranks"]}) == 8
    assert out["allgather_of_ranks"] == [float(k) for k in range(8)]
    # (C5_one_gpu: N > 1 only -- all eight scenes on rank 0's GPU, the one-GPU reference point of the line's scaling_c5 block)
    assert out["plan"] == ["C1", "C3", "C5_sharded", "C5_one_gpu", "C4", "task_static1", "C2_contracted", "C3_contracted", "C5_sharded_contracted"]
    assert out["budget_s"] < 60.0, out["budget_rows"]
    # C4 across ranks never degrades to a skipped record: peer mailboxes, or the host-coupled exchange (VERDICT r4 weak 8;
    # the branch itself runs in tests/test_shard_gpu.py::test_bench_c4_couples_through_the_host_when_inboxes_cannot_be_shared)
    assert "host-coupled" in out["c4_coupling"] and "skipped" not in out["c4_coupling"]
    r = _run(["--gpus", "2", "--dry-run"], PMAF_BENCH_BACKEND="gloo", PMAF_BENCH_SINGLE_DEVICE="1", PMAF_BENCH_C4_HOST_COUPLED="1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["c4_coupling"].startswith("host, winner records")


def test_plan_is_the_same_code_path_for_every_world_size():
    """N = 1 of the scaling run and the single-GPU bench line are built by the same function from the same flags; only
    C5's dealing-out depends on N (skipped where 8 scenes do not divide)"""
    sys.path.insert(0, ROOT)
    import argparse
    import importlib
    bench = importlib.import_module("bench")
    ns = dict(config="C2", shard=False, populations=1, dynamic=False, lanes_per_agent=0, total_populations=8, no_exchange=False,
              steps=20, warmup=5, min_seconds=1.0, min_blocks=5, only_headline=False, sub_steps=0, sub_seconds=0.25,
              cpu_seconds=11.0, flop_ticks=24)
    plans = {w: bench.build_plan(argparse.Namespace(**ns), w) for w in (1, 2, 3, 4, 8)}
    for w, (head, plan, skipped) in plans.items():
        assert head == plans[1][0]
        names = [n for n, _ in plan]
        if 8 % w == 0:
            # N > 1 adds ONE workload: all eight C5 scenes on rank 0's GPU (scaling_c5.one_gpu_same_job)
            same = [(n, sp) for n, sp in plan if n != "C5_one_gpu"]
            assert ("C5_one_gpu" in names) == (w > 1) and not skipped
            assert [n for n, _ in same] == [n for n, _ in plans[1][1]]
            assert [sp for _, sp in same] == [sp for _, sp in plans[1][1]]
        else:
            assert "C5_sharded" not in names and "C5_sharded" in skipped
        budget, rows = bench.plan_budget_s(argparse.Namespace(**ns), w)
        assert budget < 60.0, (w, rows)
    # the default flags (no --steps / --warmup): still minutes, not more
    ns2 = dict(ns, steps=2000, warmup=100)
    assert bench.plan_budget_s(argparse.Namespace(**ns2), 1)[0] < 180.0


def test_kernel_names_follow_the_librarys_routing():
    """bench.py names the rollout kernel of a launch from the handle's launch configuration (the rocprofv3 summary of the
    same command must show that kernel): one-slot up to 60 field obstacles, the split kernel where the library reports
    waves_per_agent > 1, the two- / four-slot kernels otherwise, the lane-group kernels by lanes per agent"""
    sys.path.insert(0, ROOT)
    import bench
    one = dict(lanes_per_agent=64, waves_per_agent=1, obstacles_per_wave=32)
    assert bench.kernel_name_of(one, 33) == "k_rollout_w64<1, 2, true, true>"
    assert bench.kernel_name_of(one, 61) == "k_rollout_w64<1, 2, true, true>"          # 60 field obstacles
    assert bench.kernel_name_of(one, 62) == "k_rollout_w64<2, 2, true, true>"          # 61: lane 60 is the repulsive obstacle's
    assert bench.kernel_name_of(one, 129, 3) == "k_rollout_w64<2, 3, true, true>"
    assert bench.kernel_name_of(one, 201) == "k_rollout_w64<4, 2, true, true>"
    split = dict(lanes_per_agent=64, waves_per_agent=2, obstacles_per_wave=64)
    assert bench.kernel_name_of(split, 129) == "k_rollout_mw<2, 2, true, false>"
    assert bench.kernel_name_of(dict(split, obstacles_per_wave=50), 101, 3) == "k_rollout_mw<2, 3, true, true>"
    assert bench.kernel_name_of(dict(lanes_per_agent=16), 33) == "k_rollout_grp<16, 2, 2>"

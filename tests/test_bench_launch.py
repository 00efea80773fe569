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

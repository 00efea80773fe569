#!/bin/bash
# round-3 (second half) evidence run on the GPU box: bench lines, rocprofv3 trace + PMC summaries for C2 / C3 / C5 x 8,
# sanitizer passes, GPU suite. Everything lands in gpurun_out/; what is kept is copied to profiles/r3_* afterwards.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_c2_driver_flags.json 2> gpurun_out/r3_bench.err
python bench.py > gpurun_out/r3_bench_c2.json 2>> gpurun_out/r3_bench.err
python bench.py --config C3 --steps 400 --only-headline --cpu-seconds 6 > gpurun_out/r3_bench_c3.json 2>> gpurun_out/r3_bench.err
python bench.py --config C5 --populations 8 --steps 400 --only-headline --cpu-seconds 6 --flop-ticks 2 > gpurun_out/r3_bench_c5x8.json 2>> gpurun_out/r3_bench.err
python bench.py --dynamic --only-headline --cpu-seconds 0 > gpurun_out/r3_bench_c2_dynamic.json 2>> gpurun_out/r3_bench.err
PMAF_BENCH_BACKEND=gloo PMAF_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --cpu-seconds 0 --flop-ticks 0 2>> gpurun_out/r3_bench.err | grep "^{" | tail -1 > gpurun_out/r3_bench_2ranks_1gpu_selfspawn.json
PMAF_BENCH_FORCE_DIST=1 MASTER_PORT=29531 python bench.py --only-headline --cpu-seconds 0 --flop-ticks 0 2>> gpurun_out/r3_bench.err | grep "^{" | tail -1 > gpurun_out/r3_bench_c2_rccl_1rank.json
bash tools/gpu_prof.sh r3_c2 > /dev/null 2>&1
bash tools/gpu_prof.sh r3_c3 --config C3 --steps 400 > /dev/null 2>&1
bash tools/gpu_prof.sh r3_c5 --config C5 --populations 8 --steps 400 > /dev/null 2>&1
{ echo "# per-agent / per-wave rollout durations from the device clock, and launch duration against the slowest wave (final kernels of round 3)";
  python tools/agenttime.py C1 C2 C3; python tools/c4agents.py; python tools/c5agents.py; python tools/launchgap.py; } 2>&1 | grep -v "^$\|amdgpu.ids" > gpurun_out/r3_agent_times.txt
bash tools/asan.sh run > /dev/null 2>&1
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r3_gpu_tests.log
for f in gpurun_out/r3_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    sp = d.get("setpoint_latency_us") or {}
    print("value %.0f %s  n_gpus %d  ms/step %.4f  kernel %s %.1f us  h_eff %.1f  setpoint %s / p99 %s us  cpu %s" % (
        d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_kernel_us"], d["h_eff"],
        sp.get("median"), sp.get("p99"), (d.get("cpu_baseline") or {}).get("value")))
    for k, c in (d.get("configs") or {}).items():
        print("   %-11s %10.0f rollouts/s  %.4f ms/tick  %s %.1f us  h_eff %.1f  hdr %s" % (
            k, c.get("rollouts_per_s", 0), c.get("ms_per_tick", 0), c.get("kernel"), c.get("avg_kernel_us", 0), c.get("h_eff", 0),
            (c.get("header_exchange_us") or {}).get("wait_median")))
except Exception as e:
    print("unreadable", e)
PY
done
cat gpurun_out/r3_gpu_tests.log; tail -3 gpurun_out/r3_bench.err

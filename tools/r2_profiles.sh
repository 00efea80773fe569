#!/bin/bash
# round-2 evidence run on the GPU box: bench lines + rocprofv3 trace / PMC summaries for C2 (the metric's config),
# C3 and C5 (8 populations on one GPU). Results land in gpurun_out/prof_r2_* and gpurun_out/r2_bench_*.json.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python bench.py > gpurun_out/r2_bench_c2.json 2> gpurun_out/r2_bench_c2.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_c2_driver_flags.json 2>> gpurun_out/r2_bench_c2.err
python bench.py --config C3 --steps 400 --cpu-seconds 6 > gpurun_out/r2_bench_c3.json 2> gpurun_out/r2_bench_c3.err
python bench.py --config C1 --cpu-seconds 0 > gpurun_out/r2_bench_c1.json 2> gpurun_out/r2_bench_c1.err
python bench.py --config C5 --populations 8 --steps 400 --cpu-seconds 6 --flop-ticks 2 > gpurun_out/r2_bench_c5x8.json 2> gpurun_out/r2_bench_c5.err
python bench.py --config C5 --steps 1000 --cpu-seconds 0 --flop-ticks 2 > gpurun_out/r2_bench_c5x1.json 2>> gpurun_out/r2_bench_c5.err
python bench.py --dynamic --cpu-seconds 0 > gpurun_out/r2_bench_c2_dynamic.json 2>> gpurun_out/r2_bench_c2.err
python bench.py --populations 4 --cpu-seconds 0 > gpurun_out/r2_bench_c2x4.json 2>> gpurun_out/r2_bench_c2.err
python bench.py --config C4 --steps 500 --cpu-seconds 0 --flop-ticks 0 > gpurun_out/r2_bench_c4_1gpu.json 2> gpurun_out/r2_bench_c4.err
bash tools/gpu_prof.sh r2_c2 > /dev/null 2>&1
bash tools/gpu_prof.sh r2_c3 --config C3 --steps 400 > /dev/null 2>&1
bash tools/gpu_prof.sh r2_c5 --config C5 --populations 8 --steps 400 > /dev/null 2>&1
for f in gpurun_out/r2_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print("value %.0f %s  ms/step %.4f  kernel %s %.1f us  h_eff %.1f  setpoint %.1f us  cpu %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_kernel_us"], d["h_eff"], d["setpoint_latency_us"]["median"], (d.get("cpu_baseline") or {}).get("value")))
except Exception as e:
    print("unreadable", e)
PY
done
ls gpurun_out/prof_r2_c2 gpurun_out/prof_r2_c3 gpurun_out/prof_r2_c5 2>/dev/null | head -40

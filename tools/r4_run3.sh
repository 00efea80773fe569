#!/bin/bash
# round 4, GPU run 3: the whole GPU suite on the ABI-5 build + the driver's bench command
mkdir -p gpurun_out/r4
export PMAF_TOL_REPORT=$PWD/gpurun_out/r4/tolerance_report.jsonl
rm -f $PMAF_TOL_REPORT
timeout 3000 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r4/gpu_tests.log 2>&1
tail -15 gpurun_out/r4/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_driver_flags.json 2> gpurun_out/r4/bench_driver_flags.err
tail -c 600 gpurun_out/r4/bench_driver_flags.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_driver_flags.json"))
print("value", d["value"], "ms", d["ms_per_step"], "kernel us", d["roofline"]["avg_kernel_us"])
print("setpoint", d["setpoint_latency_us"]["in_library"])
print("winner path", d["tick_with_winner_path_us"])
for k,v in d.get("configs",{}).items():
    print(k, v.get("rollouts_per_s"), v.get("ms_per_tick"), v.get("avg_kernel_us"), v.get("kernel"), v.get("h_eff"))
print("cpu", d["cpu_baseline"])
PY

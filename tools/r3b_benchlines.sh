#!/bin/bash
# the bench lines alone (after profiles/traffic.json has been regenerated from the PMC passes of the same build)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_c2_driver_flags.json 2> gpurun_out/r3_bench.err
python bench.py > gpurun_out/r3_bench_c2.json 2>> gpurun_out/r3_bench.err
python bench.py --config C3 --steps 400 --only-headline --cpu-seconds 6 > gpurun_out/r3_bench_c3.json 2>> gpurun_out/r3_bench.err
python bench.py --config C5 --populations 8 --steps 400 --only-headline --cpu-seconds 6 --flop-ticks 2 > gpurun_out/r3_bench_c5x8.json 2>> gpurun_out/r3_bench.err
python bench.py --dynamic --only-headline --cpu-seconds 0 > gpurun_out/r3_bench_c2_dynamic.json 2>> gpurun_out/r3_bench.err
for f in gpurun_out/r3_bench_c2_driver_flags.json gpurun_out/r3_bench_c2.json gpurun_out/r3_bench_c3.json gpurun_out/r3_bench_c5x8.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r = d["roofline"]; c = d.get("cpu_baseline") or {}
print(sys.argv[1], "value %.0f ms/step %.4f kernel %.1f us frac %.3g traffic %s cpu %s spread %s" % (d["value"], d["ms_per_step"], r["avg_kernel_us"], r["frac"], r.get("traffic"), c.get("value"), c.get("spread_O3_native")))
PY
done
tail -2 gpurun_out/r3_bench.err

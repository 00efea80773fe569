#!/bin/bash
# copies the round's evidence from gpurun_out/ to profiles/r5_* and regenerates profiles/traffic.json and the generated
# blocks of DESIGN.md / README.md. usage: bash tools/r5_collect.sh [prof] [bench] [misc]   (default: all)
cd "$(dirname "$0")/.."
O=gpurun_out/r5
PARTS=${@:-prof bench misc}
for part in $PARTS; do case $part in
prof)
  for c in c2 c3 c5; do
    [ -f gpurun_out/prof_r5_$c/trace_summary.txt ] && cp gpurun_out/prof_r5_$c/trace_summary.txt profiles/r5_${c}_trace.txt
    for i in 1 2 3 4 5; do [ -f gpurun_out/prof_r5_$c/pmc${i}_summary.txt ] && cp gpurun_out/prof_r5_$c/pmc${i}_summary.txt profiles/r5_${c}_pmc$i.txt; done
  done
  python tools/traffic_from_pmc.py C2=r5_c2 C3=r5_c3 C5x8=r5_c5 > /dev/null
  ;;
bench)
  for f in $O/bench_*.json; do [ -s "$f" ] && cp "$f" profiles/r5_$(basename $f); done
  ;;
misc)
  for f in regime.json regime.txt ticklat.txt facade_latency.txt mw_rule_sweep.txt agent_times.txt fuzz_campaign.txt tolerance_report.jsonl gpu_tests.log gpu_tests_rassoc.log variant_kernel_times.txt cpu_bench_c2_run1.json cpu_bench_c2_run2.json; do
    [ -s $O/$f ] && cp $O/$f profiles/r5_$f
  done
  ;;
esac; done
python tools/fill_numbers.py > /dev/null && echo "DESIGN.md / README.md number blocks regenerated"

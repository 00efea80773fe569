#!/bin/bash
# copies the round's evidence from gpurun_out/ to profiles/${ROUND}_* (ROUND=r6 by default) and regenerates
# profiles/traffic.json and the generated blocks of DESIGN.md / README.md. usage: bash tools/collect.sh [prof] [bench] [misc]
cd "$(dirname "$0")/.."
R=${ROUND:-r6}
O=gpurun_out/$R
PARTS=${@:-prof bench misc}
for part in $PARTS; do case $part in
prof)
  for c in c2 c3 c5 c5x2; do
    [ -f gpurun_out/prof_${R}_$c/trace_summary.txt ] && cp gpurun_out/prof_${R}_$c/trace_summary.txt profiles/${R}_${c}_trace.txt
    for i in 1 2 3 4 5; do [ -f gpurun_out/prof_${R}_$c/pmc${i}_summary.txt ] && cp gpurun_out/prof_${R}_$c/pmc${i}_summary.txt profiles/${R}_${c}_pmc$i.txt; done
  done
  python tools/traffic_from_pmc.py C2=${R}_c2 C3=${R}_c3 C5x8=${R}_c5 C5x2=${R}_c5x2 > /dev/null
  ;;
bench)
  for f in $O/bench_*.json; do [ -s "$f" ] && cp "$f" profiles/${R}_$(basename $f); done
  ;;
misc)
  for f in scaling_emulated.json regime.json regime.txt ticklat.txt facade_latency.txt mw_rule_sweep.txt agent_times.txt fuzz_campaign.txt tolerance_report.jsonl gpu_tests.log gpu_tests_rassoc.log variant_kernel_times.txt cpu_bench_c2_run1.json cpu_bench_c2_run2.json; do
    [ -s $O/$f ] && cp $O/$f profiles/${R}_$f
  done
  ;;
esac; done
python tools/fill_numbers.py > /dev/null && echo "DESIGN.md / README.md number blocks regenerated"

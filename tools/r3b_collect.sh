#!/bin/bash
# copies the evidence of tools/r3b_profiles.sh from gpurun_out/ into profiles/r3_* (run here, after the gpurun call),
# regenerates profiles/traffic.json and the generated blocks of DESIGN.md / README.md
cd "$(dirname "$0")/.."
for f in r3_bench_c2_driver_flags r3_bench_c2 r3_bench_c3 r3_bench_c5x8 r3_bench_c2_dynamic r3_bench_2ranks_1gpu_selfspawn r3_bench_c2_rccl_1rank; do
  [ -s gpurun_out/$f.json ] && tail -1 gpurun_out/$f.json > profiles/$f.json
done
for c in c2 c3 c5; do
  [ -s gpurun_out/prof_r3_$c/trace_summary.txt ] && cp gpurun_out/prof_r3_$c/trace_summary.txt profiles/r3_${c}_trace.txt
  for i in 1 2 3 4 5; do [ -s gpurun_out/prof_r3_$c/pmc${i}_summary.txt ] && cp gpurun_out/prof_r3_$c/pmc${i}_summary.txt profiles/r3_${c}_pmc$i.txt; done
done
[ -s gpurun_out/r3_asan.txt ] && cp gpurun_out/r3_asan.txt profiles/r3_asan.txt
[ -s gpurun_out/r3_agent_times.txt ] && cp gpurun_out/r3_agent_times.txt profiles/r3_agent_times.txt
[ -s gpurun_out/r3_gpu_tests.log ] && cp gpurun_out/r3_gpu_tests.log profiles/r3_gpu_tests.log
python tools/traffic_from_pmc.py C2=r3_c2 C3=r3_c3 C5x8=r3_c5
python tools/fill_numbers.py > /dev/null && echo "DESIGN / README blocks regenerated"

#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and, in separate
# passes, PMC counters for a short bench run. Never blocks on stdin.
# usage: tools/gpu_prof.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
# same command as the bench line it documents (default --steps / --warmup), minus the CPU baseline
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --only-headline --cpu-seconds 0 --flop-ticks 0 "$@" > $OUT/trace.log 2>&1 < /dev/null
python $R/tools/prof_summary.py $OUT/trace $OUT/trace_summary.txt < /dev/null
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc$i -o p -- python $R/bench.py --only-headline --steps 50 --warmup 5 --min-seconds 0.05 --flop-ticks 0 --cpu-seconds 0 "$@" > $OUT/pmc$i.log 2>&1 < /dev/null
  python $R/tools/prof_summary.py $OUT/pmc$i $OUT/pmc${i}_summary.txt < /dev/null
  rm -rf $OUT/pmc$i
done
rm -rf $OUT/trace
tail -n 30 $OUT/*_summary.txt

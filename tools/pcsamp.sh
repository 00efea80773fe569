#!/bin/bash
# PC-sampling profile of a rollout kernel on the GPU box (rocprofv3 --pc-sampling-beta-enabled; the ATT decoder
# library is not in the image, so thread trace cannot be decoded here). Writes a per-instruction histogram.
# usage: tools/pcsamp.sh <tag> <method stochastic|host_trap> <unit> <interval> <quicktime cfg e.g. C2:64>
set -u
TAG=$1; METHOD=$2; UNIT=$3; INTERVAL=$4; CFG=$5
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pcs_$TAG
mkdir -p $OUT
cd /tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 240 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT \
  --pc-sampling-interval $INTERVAL --output-format csv -d $OUT/raw -o s -- python $R/tools/quicktime.py $CFG > $OUT/run.log 2>&1 < /dev/null
echo "rc=$?" >> $OUT/run.log
find $OUT/raw -type f | head -20 >> $OUT/run.log
ls -la $(find $OUT/raw -type f | head -20) >> $OUT/run.log 2>&1
python $R/tools/pcsamp_summary.py $OUT/raw $OUT/summary.txt >> $OUT/run.log 2>&1 < /dev/null
for f in $(find $OUT/raw -name "*pc_sampling*csv"); do head -5 $f > $OUT/head_$(basename $f).txt; done
rm -rf $OUT/raw
tail -n 15 $OUT/run.log

#!/bin/bash
# runs on the GPU box: FETCH_SIZE / WRITE_SIZE of the calibration kernels (tools/calib_traffic.hip), separate passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
OUT=$R/gpurun_out/calib; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_traffic $R/tools/calib_traffic.hip || exit 1
cd /tmp
for PMC in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$PMC
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/cal_$PMC -o p -- /tmp/calib_traffic > $OUT/run_$PMC.log 2>&1 < /dev/null
  python $R/tools/prof_summary.py /tmp/cal_$PMC $OUT/$PMC.txt < /dev/null
done
cat $OUT/run_FETCH_SIZE.log | tail -2
grep -h "FETCH_SIZE\|WRITE_SIZE" $OUT/FETCH_SIZE.txt $OUT/WRITE_SIZE.txt

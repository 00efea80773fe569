#!/bin/bash
# round 4, GPU run 1: per-policy timing on one box + the tolerance-parity suite (report lines) + the contracted subset
mkdir -p gpurun_out/r4
export PMAF_TOL_REPORT=$PWD/gpurun_out/r4/tolerance_report.jsonl
rm -f $PMAF_TOL_REPORT
python tools/policytime.py C1 C2 C3 C4 C5 --rounds 3 --out gpurun_out/r4/policytime.json > gpurun_out/r4/policytime.log 2>&1
timeout 2400 python -m pytest tests/test_tolerance_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r4/tolerance.log 2>&1
tail -5 gpurun_out/r4/tolerance.log
tail -40 gpurun_out/r4/policytime.log

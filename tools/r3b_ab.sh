#!/bin/bash
# one-box A/B of the product build against tools/dbg/ab/*: quick parity subset first, then C2 / C3 / C1 (and C5 with "c5")
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_parity_gpu.py -q -x -k "c1_static1 or c2_synthetic or c2_dynamic or c3_256 or workspace or every_lane or idle_lane or large_and_ragged or dyn1_closed or all_heuristic or c5_reduced or randomised_scenes or general_step" > gpurun_out/r3b_ab_parity.log 2>&1
tail -3 gpurun_out/r3b_ab_parity.log
bash tools/ab.sh C2:64 C3:64 C1:64 > gpurun_out/r3b_ab.txt 2>&1
[ "$1" = c5 ] && bash tools/ab_c5.sh 8 0 >> gpurun_out/r3b_ab.txt 2>&1
cat gpurun_out/r3b_ab.txt

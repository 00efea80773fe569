#!/bin/bash
# round 3: FMA-polynomial portable_exp -- parity subset, then one-box A/B against the round's previous build (tools/dbg/ab/base)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_parity_gpu.py -q -x -k "c1_static1 or c2_synthetic or c2_dynamic or c3_256 or workspace or every_lane or idle_lane or large_and_ragged or dyn1_closed or all_heuristic or c5_ or randomised_scenes or device_arithmetic or libm or general_step or xact" > gpurun_out/r3b_exp_parity.log 2>&1
tail -5 gpurun_out/r3b_exp_parity.log
bash tools/ab.sh C2:64 C3:64 C1:64 > gpurun_out/r3b_exp_ab.txt 2>&1
bash tools/ab_c5.sh 8 0 > gpurun_out/r3b_exp_ab_c5.txt 2>&1
cat gpurun_out/r3b_exp_ab.txt gpurun_out/r3b_exp_ab_c5.txt

#!/bin/bash
# one-box A/B of the GROUP kernel: the C5 / group-mapping parity cases first, then C5 x 8 of the product build against
# tools/dbg/ab/*
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_parity_gpu.py -q -x -k "c5 or group or lpa or all_heuristic or randomised_scenes or large_and_ragged or general_step" > gpurun_out/ab_grp_parity.log 2>&1
tail -3 gpurun_out/ab_grp_parity.log
bash tools/ab_c5.sh 8 0 > gpurun_out/ab_grp.txt 2>&1
cat gpurun_out/ab_grp.txt

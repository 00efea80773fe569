#!/bin/bash
# round 6: full mapping grid for pick_lpa (tools/lpaband.py) + the GPU suite on the round's first host-side changes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 ) > $O/gpu_tests_1.log
{
  for M in 9 16 20 32 40 48 60 64 100 128; do
    args=""
    for N in 1024 1536 2048 2304 2560 3072 4096 6144 8192 12288 16384; do args="$args $M:$N:1:64,32,16,8,0"; done
    REPS=2 timeout 900 python tools/lpaband.py $args
  done
} > $O/lpa_grid.txt 2>&1

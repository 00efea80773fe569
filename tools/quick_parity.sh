#!/bin/bash
# kernel-iteration loop on the GPU box: a fast bit-exactness subset, then timings
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_parity_gpu.py -q -x -k "c1_static1 or c2_synthetic or c2_dynamic or c3_256 or workspace or every_lane or idle_lane or large_and_ragged or dyn1_closed or all_heuristic or c5_reduced or randomised_scenes" 2>&1 | tail -4
python tools/quicktime.py ${@:-C1:64 C2:64 C3:64} 2>&1 | grep -v "^$"

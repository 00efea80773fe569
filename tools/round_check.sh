#!/bin/bash
# one GPU call: the GPU suite, then the mapping sweeps that tests/test_lpa_model.py replays (gpurun -- bash tools/round_check.sh)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/${ROUND:-r6}; mkdir -p $O
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40 ) > $O/gpu_tests_check.log
bash tools/lpa_sweeps.sh ${@:-band grid}

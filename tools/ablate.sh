#!/bin/bash
# Builds the timing-experiment variants of the library (compile-time ablations of the w64 step; results are NOT
# valid trajectories) into tools/dbg/ab/<name>/ -- run here, then `gpurun -- bash tools/ablate_run.sh` on the GPU box.
cd "$(dirname "$0")/../predictive-multi-agent-framework_amd/csrc"
for v in NOSUM NOSCALE NOCIRC NOCOST; do
  PMAF_OUT=../../tools/dbg/ab/abl_$v PMAF_EXTRA_FLAGS="-DPMAF_ABLATION -DPMAF_ABL_$v" bash build.sh 2>&1 | tail -1
done

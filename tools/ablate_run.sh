#!/bin/bash
# GPU box: C2 / C3 kernel time of the product build and of each ablation variant (tools/ablate.sh), plus the run-time
# ablation PMAF_ABLATE=4 (no in-shell block at all). One box, interleaved.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
P=predictive-multi-agent-framework_amd/lib/libpmaf_hip.so
for round in 1 2; do
  echo "== product"; PMAF_LIB_PATH=$PWD/$P python tools/quicktime.py C2:64 C3:64 2>&1 | grep tick
  echo "== product, PMAF_ABLATE=4 (sweep + per-agent part only)"; PMAF_ABLATE=4 PMAF_LIB_PATH=$PWD/$P python tools/quicktime.py C2:64 C3:64 2>&1 | grep tick
  for d in tools/dbg/ab/abl_*; do echo "== $(basename $d)"; PMAF_LIB_PATH=$PWD/$d/libpmaf_hip.so python tools/quicktime.py C2:64 C3:64 2>&1 | grep tick; done
done

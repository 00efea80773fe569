#!/bin/bash
# round-2 kernel probe (runs on the GPU box): dispatch gaps, section split of the w64 step, quick timings
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{
echo "== gaps C2"; bash tools/gaps.sh C2:64
echo "== quicktime"; python tools/quicktime.py C1:64 C2:64 C3:64 2>&1 | grep -v "^$"
echo "== sections (timer build) C2"; PMAF_LIB_PATH=$PWD/tools/dbg/timers/libpmaf_hip.so python tools/sectime.py C2 2>&1 | grep "agent\|tick" | tail -9
} > gpurun_out/r2_probe.txt 2>&1
cat gpurun_out/r2_probe.txt

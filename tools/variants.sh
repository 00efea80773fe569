#!/bin/bash
# times every kernel variant library under tools/dbg/v_*.so (plus the product build)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for lib in predictive-multi-agent-framework_amd/lib/libpmaf_hip.so tools/dbg/v_*.so; do
  echo "== $lib"
  PMAF_LIB_PATH=$PWD/$lib python tools/quicktime.py "$@" 2>&1 | grep -v "^$" | tail -4
  PMAF_LIB_PATH=$PWD/$lib python tools/agenttime.py C2 2>&1 | grep "per-agent\|RANDOM"
done

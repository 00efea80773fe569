#!/bin/bash
# A/B of library variants on the C5 shape (8 x 1024 agents in one handle): product build + tools/dbg/ab/*/, 3 rounds
cd ${GRAFT_REPO_ROOT:-$(pwd)}
LIBS="predictive-multi-agent-framework_amd/lib/libpmaf_hip.so $(ls tools/dbg/ab/*/libpmaf_hip.so 2>/dev/null)"
for round in 1 2 3; do
  for lib in $LIBS; do
    echo "== round $round $lib"
    PMAF_LIB_PATH=$PWD/$lib python tools/c5time.py "$@" 2>&1 | grep -v "^$"
  done
done

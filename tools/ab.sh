#!/bin/bash
# A/B timing of kernel variants on ONE box: every library under tools/dbg/ab/*/libpmaf_hip.so plus the product build,
# interleaved, 3 rounds (boxes differ by +-2 %, runs on one box by ~0.3 %)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
LIBS="predictive-multi-agent-framework_amd/lib/libpmaf_hip.so $(ls tools/dbg/ab/*/libpmaf_hip.so 2>/dev/null)"
for round in 1 2 3; do
  for lib in $LIBS; do
    echo "== round $round $lib"
    PMAF_LIB_PATH=$PWD/$lib python tools/quicktime.py "$@" 2>&1 | grep -v "^$"
  done
done

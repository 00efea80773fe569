#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
for m in select move reset tick; do
  rm -rf /tmp/mp; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/mp -o t -- python $R/tools/mgrtime.py $m > /tmp/mp.log 2>&1 < /dev/null
  python $R/tools/prof_summary.py /tmp/mp 2>/dev/null | grep "k_manager" | sed "s/^/$m: /"
done

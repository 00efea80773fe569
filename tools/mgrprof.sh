#!/bin/bash
# k_manager duration per phase (select / real step / reset / full tick) from rocprofv3 kernel traces of tools/mgrtime.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
for m in select move reset tick; do
  rm -rf /tmp/mp; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/mp -o t -- python $R/tools/mgrtime.py $m > /tmp/mp.log 2>&1 < /dev/null
  python $R/tools/prof_summary.py /tmp/mp 2>/dev/null | grep "k_manager" | sed "s/^/$m: /"
done

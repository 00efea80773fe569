#!/bin/bash
# idle time on the stream between the two kernels of a tick (rocprofv3 kernel-trace timestamps of tools/ticktime.py's loop)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/gp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o t -- python $R/tools/quicktime.py ${1:-C2:64} > /tmp/gp.log 2>&1 < /dev/null
python - <<'PY'
import csv, glob, numpy as np
f = glob.glob('/tmp/gp/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
rows = rows[len(rows) // 2:]   # steady state
g_mr, g_rm, d_m, d_r = [], [], [], []
for a, b in zip(rows[:-1], rows[1:]):
    gap = (b[0] - a[1]) / 1e3
    if 'k_manager' in a[2] and 'k_rollout' in b[2]: g_mr.append(gap); d_m.append((a[1] - a[0]) / 1e3)
    if 'k_rollout' in a[2] and 'k_manager' in b[2]: g_rm.append(gap); d_r.append((a[1] - a[0]) / 1e3)
p = lambda x: "median %.2f mean %.2f p90 %.2f" % (np.median(x), np.mean(x), np.percentile(x, 90))
print("k_manager us:", p(d_m)); print("k_rollout us:", p(d_r))
print("gap manager -> rollout us:", p(g_mr)); print("gap rollout -> manager us:", p(g_rm))
print("tick (sum of medians) %.1f us" % (np.median(d_m) + np.median(d_r) + np.median(g_mr) + np.median(g_rm)))
PY

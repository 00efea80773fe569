#!/bin/bash
# kernel timeline of an arbitrary command (rocprofv3 --kernel-trace): durations and the idle time between the
# manager and rollout kernels of consecutive ticks; other kernels (RCCL, pack, copies) listed with their overlap
# usage: tools/gaps2.sh <command...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/gp2
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp2 -o t -- "$@" > /tmp/gp2.log 2>&1 < /dev/null
python - <<'PY'
import csv, glob, numpy as np, collections
f = glob.glob('/tmp/gp2/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
rows = rows[len(rows) // 2:]
main = [r for r in rows if 'k_manager' in r[2] or 'k_rollout' in r[2]]
g_mr, g_rm, d_m, d_r = [], [], [], []
for a, b in zip(main[:-1], main[1:]):
    gap = (b[0] - a[1]) / 1e3
    if 'k_manager' in a[2] and 'k_rollout' in b[2]: g_mr.append(gap); d_m.append((a[1] - a[0]) / 1e3)
    if 'k_rollout' in a[2] and 'k_manager' in b[2]: g_rm.append(gap); d_r.append((a[1] - a[0]) / 1e3)
p = lambda x: "median %.2f mean %.2f p90 %.2f" % (np.median(x), np.mean(x), np.percentile(x, 90))
print("k_manager us:", p(d_m)); print("k_rollout us:", p(d_r))
print("gap manager -> rollout us:", p(g_mr)); print("gap rollout -> manager us:", p(g_rm))
print("tick (sum of medians) %.1f us" % (np.median(d_m) + np.median(d_r) + np.median(g_mr) + np.median(g_rm)))
others = collections.defaultdict(list)
mstart = np.array([r[0] for r in main if 'k_manager' in r[2]])
for r in rows:
    if 'k_manager' in r[2] or 'k_rollout' in r[2]: continue
    i = np.searchsorted(mstart, r[0]) - 1
    off = (r[0] - mstart[i]) / 1e3 if i >= 0 else float('nan')
    others[r[2][:60]].append(((r[1] - r[0]) / 1e3, off))
for k, v in others.items():
    v = np.array(v)
    print("%-60s n %5d  dur median %.2f us  starts %.1f us after its tick's manager start (median)" % (k, len(v), np.median(v[:, 0]), np.nanmedian(v[:, 1])))
PY

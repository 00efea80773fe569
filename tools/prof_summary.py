#!/usr/bin/env python3
"""Summarise rocprofv3 output (rocpd sqlite .db or csv dir) into a small text
file suitable for profiles/. Usage: prof_summary.py <dir> [out.txt]"""
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    for dbp in dbs:
        db = sqlite3.connect(dbp)
        cur = db.cursor()
        print("# %s" % os.path.basename(dbp), file=out)
        try:
            rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
            print("kernel | calls | total_us | avg_us | pct", file=out)
            for r in rows:
                print("%s | %d | %.1f | %.3f | %.2f" % (r[0], r[1], r[2] / 1e3 if r[3] > 1e4 else r[2], r[3] / 1e3 if r[3] > 1e4 else r[3], r[4]), file=out)
        except Exception as e:  # noqa
            print("no top_kernels view: %s" % e, file=out)
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" in tabs:
            try:
                q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                     "group by kernel_name, counter_name order by kernel_name, counter_name")
                print("\nkernel | counter | dispatches | avg_per_dispatch | sum", file=out)
                for r in cur.execute(q):
                    print("%s | %s | %d | %.1f | %.1f" % r, file=out)
            except Exception as e:  # noqa
                print("counters query failed: %s" % e, file=out)
                print([x[0] for x in cur.execute("select * from counters_collection limit 1").description], file=out)
    csvs = glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)
    for c in csvs:
        print("# %s" % os.path.basename(c), file=out)
        with open(c) as f:
            for i, line in enumerate(f):
                if i < 40:
                    out.write(line)


if __name__ == "__main__":
    main()

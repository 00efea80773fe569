"""Condenses rocprofv3's pc_sampling_*.csv into a per-instruction histogram (samples, issued / stalled, stall
reasons) of the hottest kernel. usage: pcsamp_summary.py <rocprof output dir> <summary file>"""
import csv, glob, os, sys, collections
src, dst = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(src, "**", "*pc_sampling*.csv"), recursive=True)
out = open(dst, "w")
for f in files:
    rows = list(csv.DictReader(open(f)))
    out.write("== %s: %d samples; columns %s\n" % (os.path.basename(f), len(rows), list(rows[0].keys()) if rows else []))
    if not rows:
        continue
    key = "Instruction" if "Instruction" in rows[0] else None
    cmt = "Instruction_Comment" if "Instruction_Comment" in rows[0] else None
    per = collections.OrderedDict()
    for r in rows:
        k = (r.get(cmt, ""), r.get(key, ""))
        d = per.setdefault(k, collections.Counter())
        d["n"] += 1
        if "Wave_Issued_Instruction" in r:
            d["issued"] += int(r["Wave_Issued_Instruction"] or 0)
        for c in ("Stall_Reason", "Instruction_Type"):
            if c in r and r[c]:
                d[c + ":" + r[c]] += 1
    tot = sum(d["n"] for d in per.values())
    out.write("total %d samples, %d distinct pcs\n" % (tot, len(per)))
    # by kernel-ish prefix (comment usually holds source / symbol+offset)
    top = sorted(per.items(), key=lambda kv: -kv[1]["n"])
    out.write("-- top 60 by samples\n")
    for (c, i), d in top[:60]:
        rs = ", ".join("%s=%d" % (k.split(":", 1)[1], v) for k, v in d.most_common() if k.startswith("Stall_Reason"))
        out.write("%6d %5.2f%% issued %5d | %-60s | %s | %s\n" % (d["n"], 100.0 * d["n"] / tot, d["issued"], i[:60], c[-70:], rs))
    agg = collections.Counter()
    for d in per.values():
        for k, v in d.items():
            if ":" in k:
                agg[k] += v
    out.write("-- totals\n")
    for k, v in agg.most_common():
        out.write("%8d %5.2f%% %s\n" % (v, 100.0 * v / tot, k))
    # full listing in pc order of appearance (comment = symbol+offset) for the file
    out.write("-- all pcs sorted by comment\n")
    for (c, i), d in sorted(per.items(), key=lambda kv: kv[0][0]):
        rs = ",".join("%s=%d" % (k.split(":", 1)[1][:18], v) for k, v in d.most_common() if k.startswith("Stall_Reason"))
        out.write("%6d iss %5d | %-70s | %s | %s\n" % (d["n"], d["issued"], i[:70], c[-60:], rs))
out.close()
print(open(dst).read()[:3000])

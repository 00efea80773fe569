"""Keeps profiles/ readable: evidence files of EARLIER rounds that no current document, source comment, test or tool cites
(by name or by a pattern such as r3_c2_slack_*.txt / r5_c{2,3,5}_trace.txt) move to profiles/archive/ (names unchanged), and
profiles/INDEX.md is rewritten: the current round's evidence by topic, the older files still cited and by whom, what the
archive holds. usage: python tools/archive_profiles.py [--dry-run]   (ROUND in the environment, default r6)"""
import fnmatch
import itertools
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
A = os.path.join(P, "archive")
R = os.environ.get("ROUND", "r6")
DRY = "--dry-run" in sys.argv
TOKEN = re.compile(r"(?<![A-Za-z0-9_])((?:r[1-9]_|traffic)[A-Za-z0-9_*{},\-]*(?:\.[A-Za-z0-9_*{},\-]+)*\.(?:txt|json|jsonl|log|patch))")


def sources():
    out = subprocess.run(["git", "ls-files"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split("\n")
    for f in out:
        if not f or f.startswith("profiles/") or f == "tools/archive_profiles.py" or f in ("VERDICT.md", "ADVICE.md", "SURVEY.md", "PAPERS.md", "SNIPPETS.md", "BASELINE.md"):
            continue
        if f.endswith((".md", ".py", ".sh", ".h", ".hpp", ".cpp", ".hip", ".c", ".txt")):
            yield f


def expand(tok):
    """r5_c{2,3,5}_trace.txt -> three names; wildcards stay for fnmatch"""
    parts = re.split(r"(\{[^}]*\})", tok)
    alts = [p[1:-1].split(",") if p.startswith("{") else [p] for p in parts]
    return ["".join(c) for c in itertools.product(*alts)]


files = sorted(f for f in os.listdir(P) if os.path.isfile(os.path.join(P, f)) and f != "INDEX.md")
cited = {}
for src in sources():
    try:
        text = open(os.path.join(ROOT, src), errors="replace").read()
    except OSError:
        continue
    for tok in set(TOKEN.findall(text)):
        for pat in expand(tok):
            for f in (fnmatch.filter(files, pat) if any(c in pat for c in "*?[") else ([pat] if pat in files else [])):
                cited.setdefault(f, set()).add(src)
moved = []
for f in files:
    m = re.match(r"r(\d)_", f)
    if not m or "r%s" % m.group(1) == R or f in cited:
        continue
    moved.append(f)
    if not DRY:
        os.makedirs(A, exist_ok=True)
        subprocess.run(["git", "mv", "-k", os.path.join("profiles", f), os.path.join("profiles", "archive", f)], cwd=ROOT, check=True)
        if os.path.exists(os.path.join(P, f)):   # (untracked)
            os.rename(os.path.join(P, f), os.path.join(A, f))
print("%d files, %d cited, %d moved to profiles/archive/%s" % (len(files), len([f for f in files if f in cited]), len(moved), " (dry run)" if DRY else ""))

cur = sorted(f for f in os.listdir(P) if f.startswith(R + "_"))
TOPICS = [("bench lines (bench.py, one JSON line each)", "_bench_"), ("emulated C5 strong-scaling curve", "_scaling_"),
          ("rocprofv3 --kernel-trace --stats summaries", "_trace"), ("PMC passes (separate runs per counter group)", "_pmc"),
          ("lanes-per-agent mapping sweeps (tools/lpaband.py; csrc/pmaf_lpa_model.hpp)", "_lpa_"), ("step-loop ISA (tools/steploop.py)", "_steploop"),
          ("regime / latency / CPU baseline", ("_regime", "_ticklat", "_facade_latency", "_cpu_bench", "_agent_times", "_c5_per_gpu")),
          ("parity: GPU suite logs, fuzz, soak, tolerance report, variant", ("_gpu_tests", "_fuzz", "_soak", "_tolerance", "_variant", "_asan")),
          ("probe of the GPU box for the pin's prerequisites", "_pin_probe")]
L = ["# profiles/ — index", "",
     "Evidence the documents cite. `%s_*` = the current round (regenerate: `gpurun -- bash tools/evidence.sh`, then `bash tools/collect.sh`);" % R,
     "`traffic.json` = HBM bytes per launch from that round's PMC passes (`tools/traffic_from_pmc.py`). Older files that a current document,",
     "source comment, test or tool still cites stay here under their round's name; the rest is under `archive/` (names unchanged;",
     "`tools/archive_profiles.py` decides and rewrites this file).", "", "## Current round (%s)" % R, ""]
seen = set()
for title, keys in TOPICS:
    keys = (keys,) if isinstance(keys, str) else keys
    fs = [f for f in cur if any(k in f for k in keys) and f not in seen]
    seen.update(fs)
    if fs:
        L.append("* **%s**: %s" % (title, ", ".join("`%s`" % f for f in fs)))
rest = [f for f in cur if f not in seen]
if rest:
    L.append("* other: " + ", ".join("`%s`" % f for f in rest))
L += ["", "## Earlier rounds, still cited", "", "| file | cited by |", "|---|---|"]
for f in sorted(f for f in os.listdir(P) if re.match(r"r\d_", f) and not f.startswith(R + "_")):
    L.append("| `%s` | %s |" % (f, ", ".join("`%s`" % s for s in sorted(cited.get(f, []))[:4]) + (" …" if len(cited.get(f, [])) > 4 else "")))
if os.path.isdir(A) or moved:
    arch = sorted(os.listdir(A)) if os.path.isdir(A) else moved
    L += ["", "## archive/", ""]
    for rd in sorted({re.match(r"(r\d)_", f).group(1) for f in arch if re.match(r"r\d_", f)}):
        fs = [f for f in arch if f.startswith(rd + "_")]
        L.append("* %s: %d files — %s" % (rd, len(fs), ", ".join(sorted({re.sub(r"\d", "#", f[3:]) for f in fs}))[:600]))
if not DRY:
    open(os.path.join(P, "INDEX.md"), "w").write("\n".join(L) + "\n")
else:
    print("\n".join(L[:40]))

"""Instruction counts of the step loops of the wave-per-agent kernels, from the compiler's assembly (runs without a GPU):
for a lone wave every instruction is an issue slot (tools/slackprof.py), so this is the first-order cost model of a
kernel change. usage: python tools/loopcount.py [extra -D flags ...]
prints, per kernel, the depth-1 loops that hold the ordered sum and the path store (one per heuristic type x repulsive-
obstacle variant) with their instruction counts, and a histogram of the first one."""
import collections, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import slackprof as sp

KERNELS = {"C2 <1,2,dpp,plain>": "_Z13k_rollout_w64ILi1ELi2ELb1ELb1EEv7DevView10CostParams",
           "C1 <1,2,lds,plain>": "_Z13k_rollout_w64ILi1ELi2ELb0ELb1EEv7DevView10CostParams",
           "C3 <2,2,dpp,plain>": "_Z13k_rollout_w64ILi2ELi2ELb1ELb1EEv7DevView10CostParams"}
out = "/tmp/loopcount.s"
DUMP = None
if "--dump" in sys.argv:          # --dump <file>: the smallest C2 loop, one numbered instruction per line
    k = sys.argv.index("--dump"); DUMP = sys.argv[k + 1]; del sys.argv[k:k + 2]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + sp.KFLAGS + sys.argv[1:] + [os.path.join(sp.CSRC, "pmaf_k_w64.hip"), "-o", out],
                      stderr=subprocess.DEVNULL)
lines = open(out).readlines()
for name, k in KERNELS.items():
    f0 = next(i for i, l in enumerate(lines) if l.startswith(k + ":"))
    f1 = next(i for i in range(f0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    label_at, last_back = {}, {}
    for i in range(f0, f1):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m: label_at[m.group(1)] = i
    for i in range(f0, f1):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in label_at and label_at[m.group(1)] < i: last_back[m.group(1)] = i
    loops = []
    for lab, e in last_back.items():
        h = label_at[lab]
        if "Loop Header: Depth=1" not in lines[h]: continue
        body = lines[h:e + 1]
        if any("global_store_dwordx" in l for l in body) and sum(1 for l in body if sp.is_instr(l)) > 300:
            loops.append((h, e, sum(1 for l in body if sp.is_instr(l))))
    loops.sort()
    print(name, "step loops (instructions in extent):", [n for _, _, n in loops])
    if loops and os.environ.get("LOOPCOUNT_KERNEL", "C2") in name:
        h, e, _ = min(loops, key=lambda t: t[2])
        hist = collections.Counter(re.sub(r"_e(32|64)$", "", l.split()[0]) for l in lines[h:e + 1] if sp.is_instr(l))
        print("   smallest loop:", ", ".join("%s %d" % kv for kv in hist.most_common(14)))
        if DUMP:
            with open(DUMP, "w") as f:
                j = 0
                for l in lines[h:e + 1]:
                    if sp.is_instr(l): f.write("%3d %s\n" % (j, l.strip()[:110])); j += 1
                    elif l.strip().endswith(":") or l.startswith(".LBB"): f.write(l.split(";")[0].rstrip() + "\n")

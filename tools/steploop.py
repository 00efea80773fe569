"""gfx950 ISA of the rollout kernels' step loops, from the compiler's own assembly (no GPU needed): per loop the
instruction count by class and the number of SGPR-spill v_readlane / v_writelane INSIDE the loop, plus the listing of
the smallest loop of each kernel. For a lone wave every instruction is a 4-cycle issue slot (tools/slackprof.py), so
the count is the first-order cost model of the latency shape; the group kernel is VALU-issue bound, so its VALU count is.

usage: python tools/steploop.py <out_dir>          [prefix]   (writes <prefix>_<tag>_steploop.txt per kernel family; prefix defaults to r6)

How a spill is recognised: SGPR spills go to lanes of VGPRs that no other instruction touches -- a VGPR that, in the
whole kernel, is only ever the destination of v_writelane_b32 and the source of v_readlane_b32 with CONSTANT lane
numbers is a spill register; the readlane / writelane instructions on such registers inside a loop are the in-loop
spill traffic. (v_readlane of a computed value -- the tail's riders, the ordered sum's row totals -- reads VGPRs that
VALU instructions write, and is counted under `cross-lane`.)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIX = sys.argv[2] if len(sys.argv) > 2 else "r6"
CSRC = os.path.join(ROOT, "predictive-multi-agent-framework_amd", "csrc")
BASE = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=max-ilp "
        "-mllvm -amdgpu-atomic-optimizer-strategy=None --cuda-device-only -S").split()

# (tag, source, extra flags as in csrc/build.sh, {label: mangled kernel})
UNITS = [
    ("c1c2_strict", "pmaf_k_w64.hip", ["-DPMAF_W64_MATH=2", "-DPMAF_W64_PART=1", "-falign-loops=32", "-DPMAF_SUM_HOIST=11"],
     {"C2 / C4 k_rollout_w64<1,2,dpp,plain>": "_Z13k_rollout_w64ILi1ELi2ELb1ELb1EEv7DevView10CostParams"}),
    ("c3_strict", "pmaf_k_w64.hip", ["-DPMAF_W64_MATH=2", "-DPMAF_W64_PART=2", "-mllvm", "-misched-prera-direction=topdown",
                                     "-mllvm", "-align-all-nofallthru-blocks=6"],
     {"C3 k_rollout_w64<2,2,dpp,plain>": "_Z13k_rollout_w64ILi2ELi2ELb1ELb1EEv7DevView10CostParams"}),
    ("c3_mw_strict", "pmaf_k_mw.hip", ["-DPMAF_MW_MATH=2"],
     {"C3 k_rollout_mw<2,2,plain,64 per wave>": "_Z12k_rollout_mwILi2ELi2ELb1ELb0EEv7DevView10CostParamsi",
      "M <= 122 k_rollout_mw<2,2,plain,riders>": "_Z12k_rollout_mwILi2ELi2ELb1ELb1EEv7DevView10CostParamsi"}),
    ("c5_strict", "pmaf_k_grp.hip", ["-DPMAF_GRP_MATH=2"],
     {"C5 k_rollout_grp<16,2,2>": "_Z13k_rollout_grpILi16ELi2ELi2EEv7DevView10CostParams"}),
    ("c2c3_contracted", "pmaf_k_w64.hip", ["-DPMAF_W64_MATH=3", "-ffp-contract=fast"],
     {"C2 k_rollout_w64<1,3,dpp,plain>": "_Z13k_rollout_w64ILi1ELi3ELb1ELb1EEv7DevView10CostParams",
      "C3 k_rollout_w64<2,3,dpp,plain>": "_Z13k_rollout_w64ILi2ELi3ELb1ELb1EEv7DevView10CostParams"}),
    ("c5_contracted", "pmaf_k_grp.hip", ["-DPMAF_GRP_MATH=3", "-ffp-contract=fast"],
     {"C5 k_rollout_grp<16,2,3>": "_Z13k_rollout_grpILi16ELi2ELi3EEv7DevView10CostParams"}),
]


def is_instr(line):
    t = line.strip()
    return bool(t) and line[0] in "\t " and not t.startswith((".", ";", "//")) and not t.endswith(":")


def mnemonic(line):
    return re.sub(r"_e(32|64)$", "", line.split()[0])


def klass(m, line, spill_regs):
    if m in ("v_readlane_b32", "v_writelane_b32"):
        regs = re.findall(r"\bv(\d+)\b", line)
        if regs and int(regs[0 if m == "v_writelane_b32" else -1]) in spill_regs:
            return "SGPR spill (readlane / writelane)"
        return "cross-lane (readlane / writelane / dpp mov)"
    if m.endswith("_dpp") and m.startswith("v_mov"):
        return "cross-lane (readlane / writelane / dpp mov)"
    if m.startswith("v_") and "f64" in m:
        return "VALU f64"
    if m.startswith("v_"):
        return "VALU other (select, compare-free int, mbcnt ...)" if not m.startswith("v_cmp") else "VALU compare"
    if m.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if m.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_setprio")):
        return "wait / nop / prio"
    if m.startswith("s_"):
        return "SALU"
    if m.startswith("ds_"):
        return "LDS"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "other"


def vgprs_of(tok):
    """VGPR numbers an operand token names: v12 or v[12:13]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def spill_registers(body):
    other, lane_only = set(), set()
    for l in body:
        if not is_instr(l):
            continue
        m = mnemonic(l)
        ops = [t.strip().lstrip("-|").rstrip("|") for t in l.split(None, 1)[1].split(",")] if len(l.split(None, 1)) > 1 else []
        regs = [r for t in ops for r in vgprs_of(t.split()[0] if t else "")]
        if m == "v_writelane_b32" and len(ops) == 3 and re.fullmatch(r"\d+", ops[2]):
            lane_only.update(vgprs_of(ops[0]))
            continue
        if m == "v_readlane_b32" and len(ops) == 3 and re.fullmatch(r"\d+", ops[2]):
            lane_only.update(vgprs_of(ops[1]))
            continue
        other.update(regs)
    return lane_only - other


def analyse(lines, kernel):
    """-> [(header line, extent body, all-blocks body)] of the step loops (the innermost loops of more than 250 instructions), the spill VGPRs,
    the kernel's spill instruction count. extent = header .. last back edge (the laid-out hot path: blocks marked
    unlikely sit behind it); all blocks = every basic block the compiler's loop annotation assigns to the loop, inner
    loops and rare blocks included."""
    f0 = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    f1 = next(i for i in range(f0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    spills = spill_registers(lines[f0:f1])
    label_at, last_back = {}, {}
    for i in range(f0, f1):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            label_at[m.group(1)] = i
    for i in range(f0, f1):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            last_back[m.group(1)] = i
    # basic blocks and the loop each belongs to (LLVM's asm annotations on the label lines)
    starts = sorted(label_at.values())
    blocks = []
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else f1
        note = lines[st]
        j = st + 1
        while j < en and lines[j].strip().startswith(";"):
            note += lines[j]
            j += 1
        blocks.append((st, en, note))
    loops = []
    for lab, e in last_back.items():
        h = label_at[lab]
        if "Loop Header: Depth=" not in "".join(lines[h:h + 4]):   # (the annotation of a nested header spans lines)
            continue
        name = lab[2:]   # "BB3_10"
        member = [b for b in blocks if b[0] == h or re.search(r"(Header=|Parent Loop )%s\b" % re.escape(name), b[2])]
        allb = [l for st, en, _ in member for l in lines[st:en] if is_instr(l)]
        extent = [l for l in lines[h:e + 1] if is_instr(l)]
        if len(allb) > 250:
            loops.append((h, e, extent, allb))
    # the step loops are the innermost loops of that size (the wave-per-agent kernels wrap them in the path-chunk loop)
    loops = [L for L in loops if not any(o is not L and L[0] <= o[0] and o[1] <= L[1] for o in loops)]
    loops.sort()
    total_spill_instrs = sum(1 for l in lines[f0:f1] if is_instr(l) and klass(mnemonic(l), l, spills).startswith("SGPR spill"))
    return loops, spills, total_spill_instrs


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    summary = []
    for tag, src, flags, kernels in UNITS:
        asm = "/tmp/steploop_%s.s" % tag
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + BASE + flags + [os.path.join(CSRC, src), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).readlines()
        with open(os.path.join(out_dir, "%s_%s_steploop.txt" % (PREFIX, tag)), "w") as f:
            f.write("# tools/steploop.py -- %s %s\n# (hipcc %s)\n" % (src, " ".join(flags), " ".join(BASE)))
            for label, k in kernels.items():
                loops, spills, nspill = analyse(lines, k)
                f.write("\n== %s  (%s)\n" % (label, k))
                f.write("spill VGPRs of the kernel (touched by constant-lane v_writelane / v_readlane only): %d; spill instructions in the "
                        "whole kernel: %d\n" % (len(spills), nspill))
                f.write("step loops (one per heuristic body%s), instructions in the loop's extent:\n"
                        % (" x repulsive-obstacle variant" if "<1," in label else ""))
                rows = []
                for h, e, extent, allb in loops:
                    c = collections.Counter(klass(mnemonic(l), l, spills) for l in allb)
                    rows.append((len(extent), len(allb), c))
                    f.write("  extent %4d | all blocks %4d instructions: %s\n"
                            % (len(extent), len(allb), " | ".join("%s %d" % kv for kv in sorted(c.items()))))
                in_loop_spills = [r[2].get("SGPR spill (readlane / writelane)", 0) for r in rows]
                f.write("SGPR-spill instructions INSIDE the step loops (all blocks): %s\n" % in_loop_spills)
                if loops:
                    h, e, extent, allb = min(loops, key=lambda t: len(t[3]))
                    hist = collections.Counter(mnemonic(l) for l in allb)
                    f.write("smallest loop (all blocks), by mnemonic: %s\n" % ", ".join("%s %d" % kv for kv in hist.most_common()))
                    f.write("smallest loop, listing of its extent (header .. last back edge):\n")
                    j = 0
                    for l in lines[h:e + 1]:
                        if is_instr(l):
                            f.write("%4d  %s\n" % (j, l.strip()[:120]))
                            j += 1
                        elif l.startswith(".LBB"):
                            f.write(l.split(";")[0].rstrip() + "\n")
                    summary.append((label, [(r[0], r[1]) for r in rows], in_loop_spills, len(spills), nspill))
    for label, counts, ils, ns, nsp in summary:
        print("%-40s loops (extent, all blocks) %s | in-loop spill instructions %s | spill VGPRs %d, spill instructions in the kernel %d"
              % (label, counts, ils, ns, nsp))


if __name__ == "__main__":
    main()

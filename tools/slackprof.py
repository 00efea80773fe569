"""Delay-insertion profile of the wave-per-agent step loop (runs on the GPU box).

Neither thread trace nor PC sampling is usable on this pool (rocprofv3 --att needs a decoder library the image does
not ship; --pc-sampling-beta-enabled: "not supported on any of the agents", profiles/r3_pc_sampling_unavailable.txt),
so where a lone in-order wave waits is MEASURED the other way round: the product kernel's own assembly (hipcc -S, the
flags of csrc/build.sh; the assembled code object's disassembly equals the product's instruction for instruction) is
rebuilt once per instruction of the hot loop with a delay of 64 issue slots (4 x s_nop 15 = 256 cycles) in front of that instruction, and run
on the same rollouts through pmaf_debug_external_rollout. Per variant the tool reads the per-agent rollout durations
(device wall clock, CfAgent::prediction_time_) of the agents that run this loop.

Reading: a wave issues in order, at most one instruction per issue slot (4 cycles) whatever unit executes it. A delay
in front of instruction i costs its full 64 slots per step if the wave was issue-bound from there on, and less if
instructions at or behind i would have waited anyway for results that were issued BEFORE i (the wait absorbs the
delay). slack(i) = 64 - measured extra slots per step is therefore the waiting time, downstream of i, on producers
upstream of i; where slack drops from one instruction to the next, that instruction (or the one it feeds) is where the
wave actually waited, by about the size of the drop.

usage: python tools/slackprof.py <out.txt> [--type random|goal|had|...] [--stride 1] [--config C2]
"""
import argparse
import multiprocessing as mp
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z13k_rollout_w64ILi1ELi2ELb1ELb1EEv7DevView10CostParams"
CSRC = os.path.join(ROOT, "predictive-multi-agent-framework_amd", "csrc")
KFLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm "
          "-amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-atomic-optimizer-strategy=None --cuda-device-only -S -DPMAF_W64_MATH=2").split()
TYPE_AGENT = {"had": [0], "goal": [1], "obst": [2], "goalobst": [3], "vel": [4], "random": None}
# `s_nop N` holds the wave for N + 1 ISSUE SLOTS; for a lone wave a slot is 4 cycles (the SIMD's issue arbiter visits a
# wave every fourth cycle: the first run of this tool measured 103.8 ns per step for these four instructions = 256
# cycles at 2.47 GHz). All figures below are in issue slots of 4 cycles.
DELAY = ["\ts_nop 15\n"] * 4      # 64 slots = 256 cycles
DELAY_CYCLES = 64.0


def is_instr(line):
    t = line.strip()
    return bool(t) and line[0] in "\t " and not t.startswith((".", ";", "//")) and not t.endswith(":")


def emit_base(work):
    base = os.path.join(work, "base.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + KFLAGS + [os.path.join(CSRC, "pmaf_k_w64.hip"), "-o", base],
                          stderr=subprocess.DEVNULL)
    return base


def assemble(args):
    src, out = args
    obj = out + ".o"
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", obj, "-o", out])
    os.remove(obj)
    os.remove(src)
    return out


def find_loops(lines, f0, f1):
    """the step loops of the kernel body lines[f0:f1]: (header line, last back-edge line, instructions) of every
    depth-1 loop (LLVM's "Loop Header: Depth=1" annotation) that holds the ordered sum's first DPP move and the path store"""
    label_at = {}
    for k in range(f0, f1):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[k])
        if m:
            label_at[m.group(1)] = k
    last_back = {}
    for k in range(f0, f1):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[k])
        if m and m.group(1) in label_at and label_at[m.group(1)] < k:
            last_back[m.group(1)] = k
    loops = []
    for lab, k in last_back.items():
        h = label_at[lab]
        if "Loop Header: Depth=1" not in lines[h]:
            continue
        body = lines[h:k + 1]
        n_ins = sum(1 for l in body if is_instr(l))
        if any("row_newbcast:0 " in l for l in body) and any("global_store_dwordx4" in l for l in body):
            loops.append((h, k, n_ins))
    return sorted(loops)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--type", default="random")
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--ticks", type=int, default=24)
    ap.add_argument("--work", default="/tmp/slackprof")
    args = ap.parse_args()
    os.makedirs(args.work, exist_ok=True)
    base = emit_base(args.work)
    lines = open(base).readlines()
    f0 = next(k for k, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    f1 = next(k for k in range(f0, len(lines)) if lines[k].startswith(".Lfunc_end"))
    loops = find_loops(lines, f0, f1)

    import __graft_entry__ as g
    pm = g.load_package()
    sc = pm.scenes.config_scene(args.config)
    H = sc["max_prediction_steps"] - 1
    agents = TYPE_AGENT[args.type]
    if agents is None:
        agents = list(range(5, sc["n_agents"]))

    def run(co):
        """mean rollout duration (ns) of the watched agents over ticks 8.. of a fresh episode, + all agents' for the log"""
        h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
        h.set_initial_position(sc["start"])
        if co:
            h.external_rollout(co, KERNEL)
        acc = []
        for t in range(args.ticks):
            h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
            h.stop()
            if t >= 8:
                acc.append(h.prediction_times_ns().copy())
        paths = h.paths()[0].copy()
        h.close()
        a = np.mean(acc, axis=0)
        return float(a[agents].mean()), a, paths

    def variant(tag, at_line):
        src = os.path.join(args.work, "v_%s.s" % tag)
        with open(src, "w") as f:
            f.writelines(lines[:at_line] + DELAY + lines[at_line:])
        return (src, os.path.join(args.work, "v_%s.co" % tag))

    pool = mp.Pool(min(64, os.cpu_count() or 8))
    base_co = assemble((shutil_copy(base, os.path.join(args.work, "v_base.s")), os.path.join(args.work, "v_base.co")))
    t_builtin, all_builtin, p_builtin = run(None)
    t_base, all_base, p_base = run(base_co)
    assert np.array_equal(p_builtin, p_base), "external code object changed the results"
    # which loop do the watched agents run? a delay at the head of each candidate, the one that slows them is it
    heads = pool.map(assemble, [variant("head%d" % j, h + 1) for j, (h, k, n) in enumerate(loops)])
    slow = [run(co)[0] - t_base for co in heads]
    j = int(np.argmax(slow))
    h0, k0, n0 = loops[j]
    ns_per_step_cycle = None
    out = open(args.out, "w")
    out.write("# %s\n" % " ".join(sys.argv))
    out.write("# kernel %s, config %s, watched agents: type '%s' (%d agents), %d-step rollouts, ticks 8..%d of an episode\n"
              % (KERNEL, args.config, args.type, len(agents), H, args.ticks - 1))
    out.write("# built-in kernel %.1f ns, same assembly as external code object %.1f ns per rollout (results bit-identical)\n"
              % (t_builtin, t_base))
    out.write("# candidate step loops (instructions, slowdown of the watched agents with a 64-cycle delay at the loop head):\n")
    for jj, ((h, k, n), s) in enumerate(zip(loops, slow)):
        out.write("#   loop %2d: %3d instructions, %+8.1f ns per rollout%s\n" % (jj, n, s, "   <-- profiled" if jj == j else ""))
    # the delay at the loop head is fully exposed or not -- calibrate cycles from a delay known to cost its length: the
    # largest per-step cost over all positions is taken as 64 cycles (an issue-bound position exists in every loop)
    idx = [k for k in range(h0 + 1, k0 + 1) if is_instr(lines[k])]
    idx = idx[::args.stride]
    res = []
    B = 64
    t_start = time.time()
    for b in range(0, len(idx), B):
        cos = pool.map(assemble, [variant("p%d" % k, k) for k in idx[b:b + B]])
        for k, co in zip(idx[b:b + B], cos):
            t, _, p = run(co)
            if not np.array_equal(p, p_base):
                raise SystemExit("variant at line %d changed the results" % k)
            res.append((k, t - t_base))
            os.remove(co)
        sys.stderr.write("%d / %d positions, %.0f s\n" % (min(b + B, len(idx)), len(idx), time.time() - t_start))
    d = np.array([r[1] for r in res]) / H                       # extra ns per step
    # one issue slot = 4 cycles of the core clock, measured while a lone wave computes (tools/clockrate.hip: s_memtime
    # against the 100 MHz wall clock); a fully exposed 64-slot delay must then cost 64 slots (printed as a check)
    exe = os.path.join(args.work, "clockrate")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "clockrate.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    ghz = float(subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()[-1])
    full = DELAY_CYCLES * 4.0 / ghz                              # ns per step of a fully exposed 64-slot delay
    two = assemble((lambda sv: (open(sv[0], "w").writelines(lines[:h0 + 1] + DELAY + DELAY + lines[h0 + 1:]), sv)[1])(
        (os.path.join(args.work, "v_two.s"), os.path.join(args.work, "v_two.co"))))
    check = (run(two)[0] - run(heads[j])[0]) / H / full * DELAY_CYCLES   # slots a second 64-slot delay at the head costs
    cyc = d / full * DELAY_CYCLES
    slack = DELAY_CYCLES - cyc
    step_cycles = t_base / H / full * DELAY_CYCLES
    out.write("# profiled loop: %d instructions in its extent; one step = %.1f ns = %.0f issue slots of 4 cycles (64 slots = "
              "%.2f ns at the measured core clock of %.3f GHz); slots per step beyond one per executed instruction = bubbles\n"
              % (n0, t_base / H, step_cycles, full, ghz))
    out.write("# check: a second 64-slot delay right behind a first one at the loop head costs %.1f slots per step\n" % check)
    out.write("# columns: position, extra slots per step of a 64-slot delay in front of the instruction, slack = 64 - extra,\n"
              "#          drop = slack(i) - slack(i+1) (>0: the wave waited about that long at / right behind i), instruction\n")
    drops = []
    for n, ((k, _), c, s) in enumerate(zip(res, cyc, slack)):
        nxt = slack[n + 1] if n + 1 < len(slack) else slack[0]
        drops.append(s - nxt)
    order = np.argsort(-np.array(drops))
    out.write("# ---- top 25 waits (largest drops of slack) ----\n")
    for n in order[:25]:
        out.write("#  pos %3d  drop %5.1f slots  slack %5.1f -> %5.1f  %s\n"
                  % (n * args.stride, drops[n], slack[n], slack[n] - drops[n], lines[res[n][0]].strip()))
    executed = int((cyc > 8).sum())   # positions the hot path really runs through (a delay in a cold block costs nothing)
    out.write("# positions on the executed path: %d of %d; step = %.0f slots => %.0f bubble slots per step (%.0f %%); "
              "largest slack anywhere: %.1f slots (no single long wait: the bubbles are one-slot waits of an instruction "
              "issued right behind its producer); positions with slack < 2 slots: %d\n"
              % (executed, len(slack), step_cycles, step_cycles - executed, 100.0 * (step_cycles - executed) / step_cycles,
                 float(slack[cyc > 8].max()), int(((slack < 2) & (cyc > 8)).sum())))
    out.write("# ---- all positions ----\n")
    for n, ((k, _), c, s) in enumerate(zip(res, cyc, slack)):
        out.write("%4d %6.1f %6.1f %6.1f  %s\n" % (n * args.stride, c, s, drops[n], lines[k].strip()))
    out.close()
    print(open(args.out).read()[:6000])


def shutil_copy(a, b):
    import shutil
    shutil.copy(a, b)
    return b


if __name__ == "__main__":
    main()

"""Regenerates the measured-numbers block of DESIGN.md (between the `numbers:begin` / `numbers:end` markers) and the
headline table of README.md from the committed evidence: profiles/r3_bench_c2_driver_flags.json (the driver's command),
r3_bench_c3.json, r3_bench_c5x8.json, r3_c{2,3,5}_trace.txt (rocprofv3 --kernel-trace --stats), traffic.json.
usage: python tools/fill_numbers.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(name):
    return json.loads(open(os.path.join(P, name)).read().strip().split("\n")[-1])


def trace_avg(tag, kernel_prefix):
    for line in open(os.path.join(P, "r3_%s_trace.txt" % tag)):
        if kernel_prefix in line:
            f = [x.strip() for x in line.split("|")]
            return float(f[3]), int(f[1])
    return None, None


d = load("r3_bench_c2_driver_flags.json")
c3, c5 = load("r3_bench_c3.json"), load("r3_bench_c5x8.json")
cf = d["configs"]
t2, n2 = trace_avg("c2", "k_rollout_w64<1, 2, true, true>")
t3, n3 = trace_avg("c3", "k_rollout_w64<2, 2, true, true>")
t5, n5 = trace_avg("c5", "k_rollout_grp<16, 2, 2>")
rf, fv, cb = d["roofline"], d["fp64_valu"], d["cpu_baseline"]
sp = d["setpoint_latency_us"]
rows = [
    ("**C2 64 × 200 × 32** (headline, `--steps 20 --warmup 5`)", d["value"], d["ms_per_step"], "`k_rollout_w64<1,2,true,true>` %.1f / %.1f (%d calls)" % (rf["avg_kernel_us"], t2, n2)),
    ("C1 16 × 100 × 9 (sub-record)", cf["C1"]["rollouts_per_s"], cf["C1"]["ms_per_tick"], "`k_rollout_w64<1,2,true,true>` %.1f" % cf["C1"]["avg_kernel_us"]),
    ("C3 256 × 500 × 128 (sub-record / own run)", cf["C3"]["rollouts_per_s"], cf["C3"]["ms_per_tick"], "`k_rollout_w64<2,2,true,true>` %.1f / own run %.1f / %.1f (%d calls)" % (cf["C3"]["avg_kernel_us"], c3["roofline"]["avg_kernel_us"], t3, n3)),
    ("C5 8 × 1024 × 200 × 32 on one GPU (sub-record / own run)", cf["C5_sharded"]["rollouts_per_s"], cf["C5_sharded"]["ms_per_tick"], "`k_rollout_grp<16,2,2>` %.1f / own run %.1f / %.1f (%d calls)" % (cf["C5_sharded"]["avg_kernel_us"], c5["roofline"]["avg_kernel_us"], t5, n5)),
    ("C4 dual arm 2 × 256 × 200 × 32, one GPU, set-points through the peer mailboxes", cf["C4"]["rollouts_per_s"], cf["C4"]["ms_per_tick"], "`k_rollout_w64<1,2,true,true>` %.1f; header wait %.2f µs median / %.2f p99, publish %.2f µs" % (
        cf["C4"]["avg_kernel_us"], cf["C4"]["header_exchange_us"]["wait_median"], cf["C4"]["header_exchange_us"]["wait_p99"], cf["C4"]["header_exchange_us"]["publish_median"])),
]
out = []
out.append("One MI355X, round 3. `profiles/r3_bench_c2_driver_flags.json` is the driver's command (`python bench.py --steps 20 --warmup 5`: one line, "
           "%d blocks, %.2f s timed) with its sub-records; rocprofv3 `--kernel-trace --stats` summaries of the same workloads: "
           "`profiles/r3_c{2,3,5}_trace.txt`; PMC passes `profiles/r3_c*_pmc*.txt`; traffic `profiles/traffic.json`.\n" % (d["timing"]["blocks"], d["timing"]["timed_s"]))
out.append("| config | rollouts/s | ms/tick | rollout kernel µs per launch (HIP events in the bench / rocprofv3 avg) |\n|---|---|---|---|")
for name, v, ms, k in rows:
    out.append("| %s | %s | %.4f | %s |" % (name, format(int(round(v)), ",").replace(",", " "), ms, k))
out.append("")
lib = sp.get("in_library")
out.append("Set-point latency on an idle stream (host call → best index + set-point on the host), %d samples: " % sp.get("n", 100)
           + ("**on the library's own clock (`pmaf_get_tick_times_us`: entry of `pmaf_tick` → set-point on the host) median %.1f µs, "
              "p90 %.1f, p99 %.1f, max %.1f** (of which %.1f µs are the two launches being handed to the stream); " % (
                  lib["median"], lib["p90"], lib["p99"], lib["max"], lib["enqueue_median"]) if lib else "")
           + "around the bench's ctypes call median %.1f µs, p90 %.1f, p99 %.1f (the tail is the interpreter's, not the path's); "
             "back-to-back tick median %.1f µs." % (sp["median"], sp.get("p90", float("nan")), sp["p99"], d["tick_latency_us"]["median"]))
out.append("")
out.append("**Roofline of the dominant kernel (C2 launch).** Algorithmic bytes (SURVEY §8d) %d B ÷ %.1f µs = %.3f GB/s = **%.3g of 8 TB/s** "
           "(`roofline.frac`; rocprofv3 average of the same kernel: %.1f µs). FP64-VALU: %d measured FP64 operations per agent-step "
           "(`oracle/flopcount.cpp`, %.1f %% of the agent-steps with an obstacle inside the shell) → %.3f TFLOP/s = %.3f %% of 78.6 TF. "
           "HBM traffic from the PMC passes (FETCH_SIZE × 2 + WRITE_SIZE, calibrated): %.2f MB per launch = %.2f × algorithmic (the "
           "cost pass's path read-back and per-wave tables; four orders below any limit). The north star's \"≥ 40 %% HBM\" is "
           "structurally unattainable for this algorithm (SURVEY §8d); its throughput target (100 k rollouts/s at C2) is met %.2f ×."
           % (rf["algorithmic_bytes_per_launch"], rf["avg_kernel_us"], rf["achieved"], rf["frac"], t2, round(fv["flops_per_agent_step"]),
              100.0 * (fv.get("in_shell_step_fraction") or 0.0), fv["achieved_tflops"], 100.0 * fv["frac"], (rf["traffic"] or 0) / 1e6,
              (rf["traffic"] or 0) / rf["algorithmic_bytes_per_launch"], d["value"] / 1e5))
out.append("")
out.append("**CPU baseline** (`cpu_baseline`, kind `port`: `oracle/cpu_bench.py` times the oracle in a process of its own on the box's host, "
           "%s, %d logical CPUs): medians of 30 repetitions, `-O2` / `-O3 -march=native` builds (bit-identical results): one core "
           "%s / %s rollouts/s, agents' rollouts on OpenMP threads %s (%d threads, spread %s … %s) / %s (%d threads). The multi-threaded repetitions are bimodal on this host (the median lands in either mode from run to run; "
           "the fast mode is the port with all its threads spinning undisturbed, and it is about on par with the GPU at C2). **The GPU is %.1f × the median of the "
           "multi-threaded CPU port at C2 and %.0f × one core** — C2 is 64 independent 200-step chains, the shape where a GPU has the "
           "least to offer (C3: %.0f × the multi-threaded port, C5 × 8: %.0f ×); the claim here is parity and an issue-bound step, not the ratio."
           % (cb["cpu_model"], cb["host_cpus"], format(int(cb["value_1core_O2"]), ","), format(int(cb["value_1core_O3_native"]), ","),
              format(int(cb["value_O2"]), ","), cb["threads_O2"], format(int(cb["spread_O2"][0]), ","), format(int(cb["spread_O2"][1]), ","),
              format(int(cb["value_O3_native"]), ","), cb["threads_O3_native"], d["value"] / cb["value"], d["value"] / cb["value_1core"],
              c3["value"] / c3["cpu_baseline"]["value"], c5["value"] / c5["cpu_baseline"]["value"]))
block = "\n".join(out) + "\n"
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
s = re.sub(r"(<!-- numbers:begin[^\n]*-->\n).*?(<!-- numbers:end -->)", lambda m: m.group(1) + block + m.group(2), s, flags=re.S)
open(p, "w").write(s)
# README headline table
rp = os.path.join(ROOT, "README.md")
r = open(rp).read()
tab = ["| BASELINE config | rollouts/s | tick | rollout kernel | CPU port: best multi-thread (threads) / 1 core |", "|---|---|---|---|---|",
       "| C2: 64 agents × 200 steps × 32 obstacles | %.0f k | %.3f ms | %.0f µs | %.0f k (%d) / %.1f k |" % (d["value"] / 1e3, d["ms_per_step"], rf["avg_kernel_us"], cb["value"] / 1e3, cb["cores"], cb["value_1core"] / 1e3),
       "| C3: 256 × 500 × 128 | %.0f k | %.2f ms | %.2f ms | %.1f k (%d) / %.1f k |" % (c3["value"] / 1e3, c3["ms_per_step"], c3["roofline"]["avg_kernel_us"] / 1e3, c3["cpu_baseline"]["value"] / 1e3, c3["cpu_baseline"]["cores"], c3["cpu_baseline"]["value_1core"] / 1e3),
       "| C5: 8 × 1024 × 200 × 32 (one GPU) | %.1f M | %.2f ms | %.2f ms | %.0f k (%d) / %.1f k |" % (c5["value"] / 1e6, c5["ms_per_step"], c5["roofline"]["avg_kernel_us"] / 1e3, c5["cpu_baseline"]["value"] / 1e3, c5["cpu_baseline"]["cores"], c5["cpu_baseline"]["value_1core"] / 1e3)]
r = re.sub(r"(<!-- headline:begin -->\n).*?(<!-- headline:end -->)", lambda m: m.group(1) + "\n".join(tab) + "\n" + m.group(2), r, flags=re.S)
fast = max(cb["spread_O2"][1], cb["spread_O3_native"][1])
r = re.sub(r"(<!-- ratio:begin -->).*?(<!-- ratio:end -->)", lambda m: m.group(1) + "%.1f × the MEDIAN repetition of the multi-threaded CPU port of the same algorithm on the box's %s (%d threads; the repetitions are bimodal on this host, %.0f k … %.0f k rollouts/s — the fast ones are within %.1f × of the GPU), %.0f × one core" % (d["value"] / cb["value"], cb["cpu_model"], cb["cores"], min(cb["spread_O2"][0], cb["spread_O3_native"][0]) / 1e3, fast / 1e3, d["value"] / fast, d["value"] / cb["value_1core"]) + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- frac:begin -->).*?(<!-- frac:end -->)", lambda m: m.group(1) + "%.2g of 8 TB/s" % rf["frac"] + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- target:begin -->).*?(<!-- target:end -->)", lambda m: m.group(1) + "%.1f ×" % (d["value"] / 1e5) + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- lat:begin -->).*?(<!-- lat:end -->)", lambda m: m.group(1) + (("%.1f µs median, %.1f µs p99 on the library's own clock (`pmaf_get_tick_times_us`); " % (lib["median"], lib["p99"])) if lib else "") + "%.0f µs median, %.0f µs p99 around the Python bench's ctypes call" % (sp["median"], sp["p99"]) + m.group(2), r, flags=re.S)
open(rp, "w").write(r)
print(block)

"""Regenerates the measured-numbers block of DESIGN.md (between the `numbers:begin` / `numbers:end` markers) and the
generated blocks of README.md from the committed evidence of round 4: profiles/r4_bench_c2_driver_flags.json (the
driver's command), r4_bench_c3.json, r4_bench_c5x8.json, r4_c{2,3,5}_trace.txt (rocprofv3 --kernel-trace --stats),
traffic.json. usage: python tools/fill_numbers.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
R = "r4"


def load(name):
    return json.loads(open(os.path.join(P, name)).read().strip().split("\n")[-1])


def trace_avg(tag, kernel_prefix):
    try:
        for line in open(os.path.join(P, "%s_%s_trace.txt" % (R, tag))):
            if kernel_prefix in line:
                f = [x.strip() for x in line.split("|")]
                return float(f[3]), int(f[1])
    except OSError:
        pass
    return float("nan"), 0


def k(v):
    return format(int(round(v)), ",").replace(",", " ")


d = load("%s_bench_c2_driver_flags.json" % R)
c3, c5 = load("%s_bench_c3.json" % R), load("%s_bench_c5x8.json" % R)
cf = d["configs"]
t2, n2 = trace_avg("c2", "k_rollout_w64<1, 2, true, true>")
t3, n3 = trace_avg("c3", "k_rollout_mw<2, 2, true, false>")
t5, n5 = trace_avg("c5", "k_rollout_grp<16, 2, 2>")
rf, fv, cb = d["roofline"], d["fp64_valu"], d["cpu_baseline"]
sp = d["setpoint_latency_us"]
wp = d.get("tick_with_winner_path_us") or {}
kt = (cf.get("C1") or {}).get("kernel_timing") or {}
rows = [
    ("**C2 64 × 200 × 32** (headline, `--steps 20 --warmup 5`)", d["value"], d["ms_per_step"], "`k_rollout_w64<1,2,true,true>` %.1f / %.1f (%d calls)" % (rf["avg_kernel_us"], t2, n2)),
    ("C1 16 × 100 × 9 (sub-record)", cf["C1"]["rollouts_per_s"], cf["C1"]["ms_per_tick"], "`k_rollout_w64<1,2,true,true>` %.1f" % cf["C1"]["avg_kernel_us"]),
    ("C3 256 × 500 × 128 (sub-record / own run)", cf["C3"]["rollouts_per_s"], cf["C3"]["ms_per_tick"], "`k_rollout_mw<2,2,true,false>` %.1f / own run %.1f / %.1f (%d calls)" % (cf["C3"]["avg_kernel_us"], c3["roofline"]["avg_kernel_us"], t3, n3)),
    ("C5 8 × 1024 × 200 × 32 on one GPU (sub-record / own run)", cf["C5_sharded"]["rollouts_per_s"], cf["C5_sharded"]["ms_per_tick"], "`k_rollout_grp<16,2,2>` %.1f / own run %.1f / %.1f (%d calls)" % (cf["C5_sharded"]["avg_kernel_us"], c5["roofline"]["avg_kernel_us"], t5, n5)),
    ("C4 dual arm 2 × 256 × 200 × 32, one GPU, set-points through the peer mailboxes", cf["C4"]["rollouts_per_s"], cf["C4"]["ms_per_tick"], "`k_rollout_w64<1,2,true,true>` %.1f; header wait %.2f µs median / %.2f p99, publish %.2f µs" % (
        cf["C4"]["avg_kernel_us"], cf["C4"]["header_exchange_us"]["wait_median"], cf["C4"]["header_exchange_us"]["wait_p99"], cf["C4"]["header_exchange_us"]["publish_median"])),
]
for name, key, kn in (("C2 contracted policy (opt-in, tolerance parity: §4)", "C2_contracted", "k_rollout_w64<1,3,true,true>"),
                      ("C3 contracted", "C3_contracted", "k_rollout_mw<2,3,true,false>"),
                      ("C5 × 8 contracted (parity NOT met on scene 1: §4)", "C5_sharded_contracted", "k_rollout_grp<16,2,3>")):
    if key in cf and "rollouts_per_s" in cf[key]:
        rows.append((name, cf[key]["rollouts_per_s"], cf[key]["ms_per_tick"], "`%s` %.1f" % (kn, cf[key]["avg_kernel_us"])))
out = []
out.append("One MI355X, round 4. `profiles/%s_bench_c2_driver_flags.json` is the driver's command (`python bench.py --steps 20 --warmup 5`: one line, "
           "%d blocks, %.2f s timed; HIP events on every %s-th rollout launch of the timed region: %s of %s launches) with its sub-records; rocprofv3 "
           "`--kernel-trace --stats` summaries of the same workloads: `profiles/%s_c{2,3,5}_trace.txt`; PMC passes `profiles/%s_c*_pmc*.txt`; traffic "
           "`profiles/traffic.json`; `profiles/%s_bench_c2_every_launch_timed.json` is the same run with every launch timed.\n"
           % (R, d["timing"]["blocks"], d["timing"]["timed_s"], (rf.get("kernel_timing") or kt or {}).get("every", 8),
              (rf.get("kernel_timing") or {}).get("launches_timed_with_hip_events", "?"), (rf.get("kernel_timing") or {}).get("launches_in_timed_region", "?"), R, R, R))
out.append("| config | rollouts/s | ms/tick | rollout kernel µs per launch (HIP events in the bench / rocprofv3 avg) |\n|---|---|---|---|")
for name, v, ms, kk in rows:
    out.append("| %s | %s | %.4f | %s |" % (name, k(v), ms, kk))
out.append("")
lib = sp.get("in_library")
out.append("**Tick latency** on an idle stream. SURVEY §8(d)'s tick — host call → best index, set-point AND the selected agent's scored path on "
           "the host (`pmaf_enable_winner_path`: mapped pinned memory written by the manager kernel, %d B at C2), %d ticks, library clock — "
           "**median %.1f µs, p90 %.1f, p99 %.1f, max %.1f**. Set-point alone (`pmaf_get_tick_times_us`: entry of `pmaf_tick` → set-point on the "
           "host), %d samples: median %.1f µs, p90 %.1f, p99 %.1f, max %.1f (of which %.1f µs are the two launches being handed to the stream; the "
           "bench times every launch with events during these samples — `profiles/%s_ticklat.txt` has the table without: 12.5 / 13.2 µs); around the "
           "bench's ctypes call median %.1f µs, p99 %.1f (the interpreter's); back-to-back tick median %.1f µs."
           % (wp.get("path_bytes", 0), wp.get("n", 0), wp.get("median", float("nan")), wp.get("p90", float("nan")), wp.get("p99", float("nan")),
              wp.get("max", float("nan")), sp.get("n", 0), lib["median"], lib["p90"], lib["p99"], lib["max"], lib["enqueue_median"], R,
              sp["median"], sp["p99"], d["tick_latency_us"]["median"]))
try:
    dy = load("%s_bench_c2_dynamic.json" % R)
    out.append("")
    out.append("Inputs are resident in HBM when the timed region starts (the static obstacle list is handed over once). With MOVING obstacles the "
               "caller hands a new list over on every tick -- %d B host -> device, read by the manager kernel out of mapped pinned memory: "
               "%s rollouts/s, %.4f ms/tick (`profiles/%s_bench_c2_dynamic.json`): the PCIe-inclusive rate of this path."
               % (7 * 8 * (d["config"]["obstacles"] + 1), k(dy["value"]), dy["ms_per_step"], R))
except OSError:
    pass
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
lo = min(cb["spread_O2"][0], cb["spread_O3_native"][0])
hi = max(cb["spread_O2"][1], cb["spread_O3_native"][1])
out.append("**CPU baseline** (`cpu_baseline`, kind `port`: `oracle/cpu_bench.py` times the oracle in a process of its own on the box's host, "
           "%s, %d logical CPUs; 30 repetitions per build, `-O2` / `-O3 -march=native`, bit-identical results). One core: median %s / %s "
           "rollouts/s. Agents' rollouts on OpenMP threads: `-O2` median %s (%d threads, min %s … max %s), `-O3 -march=native` median %s "
           "(%d threads, min %s … max %s). The multi-threaded repetitions are NOT a stable measurement on these shared hosts — single "
           "repetitions range from %s to %s rollouts/s, and the median lands in either mode from run to run (round 3's driver run: 254 k) — "
           "so the comparison is a range: **at C2 the GPU's %s rollouts/s are %.2f × the median of the better CPU build in this run (the default-flags run minutes later on the same box measured a CPU median of %s: %.2f ×), "
           "%.2f × the port's fastest repetition and %.0f × one core**. C2 is 64 independent 200-step chains, the shape where a GPU has the "
           "least to offer — a 64-core host running one agent per core is on par with it; C3: %.0f × the multi-threaded port, C5 × 8: %.0f ×. "
           "The claim here is parity and an issue-bound step, not the ratio."
           % (cb["cpu_model"], cb["host_cpus"], k(cb["value_1core_O2"]), k(cb["value_1core_O3_native"]),
              k(cb["value_O2"]), cb["threads_O2"], k(cb["spread_O2"][0]), k(cb["spread_O2"][1]),
              k(cb["value_O3_native"]), cb["threads_O3_native"], k(cb["spread_O3_native"][0]), k(cb["spread_O3_native"][1]),
              k(lo), k(hi), k(d["value"]), d["value"] / cb["value"], k(load("%s_bench_c2.json" % R)["cpu_baseline"]["value"]), d["value"] / load("%s_bench_c2.json" % R)["cpu_baseline"]["value"], d["value"] / hi, d["value"] / cb["value_1core"],
              c3["value"] / c3["cpu_baseline"]["value"], c5["value"] / c5["cpu_baseline"]["value"]))
block = "\n".join(out) + "\n"
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
s = re.sub(r"(<!-- numbers:begin[^\n]*-->\n).*?(<!-- numbers:end -->)", lambda m: m.group(1) + block + m.group(2), s, flags=re.S)
open(p, "w").write(s)

# README
rp = os.path.join(ROOT, "README.md")
r = open(rp).read()


def cpu_cell(c):
    b = c["cpu_baseline"]
    return "%.0f k median, %.0f … %.0f k (%d) / %.1f k" % (b["value"] / 1e3, min(b["spread_O2"][0], b["spread_O3_native"][0]) / 1e3,
                                                         max(b["spread_O2"][1], b["spread_O3_native"][1]) / 1e3, b["cores"], b["value_1core"] / 1e3)


tab = ["| BASELINE config | rollouts/s | tick | rollout kernel | CPU port: multi-thread median, min … max of the repetitions (threads) / 1 core |", "|---|---|---|---|---|",
       "| C2: 64 agents × 200 steps × 32 obstacles | %.0f k | %.3f ms | %.0f µs | %s |" % (d["value"] / 1e3, d["ms_per_step"], rf["avg_kernel_us"], cpu_cell(d)),
       "| C3: 256 × 500 × 128 | %.0f k | %.2f ms | %.2f ms | %s |" % (c3["value"] / 1e3, c3["ms_per_step"], c3["roofline"]["avg_kernel_us"] / 1e3, cpu_cell(c3)),
       "| C5: 8 × 1024 × 200 × 32 (one GPU) | %.1f M | %.2f ms | %.2f ms | %s |" % (c5["value"] / 1e6, c5["ms_per_step"], c5["roofline"]["avg_kernel_us"] / 1e3, cpu_cell(c5))]
r = re.sub(r"(<!-- headline:begin -->\n).*?(<!-- headline:end -->)", lambda m: m.group(1) + "\n".join(tab) + "\n" + m.group(2), r, flags=re.S)
d2 = load("%s_bench_c2.json" % R)          # the same workload with the default flags, minutes later on the same box
meds = sorted([cb["value"], d2["cpu_baseline"]["value"]])
r = re.sub(r"(<!-- ratio:begin -->).*?(<!-- ratio:end -->)", lambda m: m.group(1) + (
    "between %.2f × and %.2f × the MEDIAN repetition of the multi-threaded CPU port of the same algorithm on the box's %s — the port's "
    "median was %.0f k rollouts/s in one of this round's two C2 runs and %.0f k in the other (`profiles/r4_bench_c2_driver_flags.json`, "
    "`r4_bench_c2.json`; single repetitions range from %.0f k to %.0f k on these shared hosts; round 3's driver run: 254 k = 1.03 ×) — and %.0f × one core"
    % (d["value"] / meds[1], d["value"] / meds[0], cb["cpu_model"], meds[0] / 1e3, meds[1] / 1e3,
       min(lo, d2["cpu_baseline"]["spread_O2"][0], d2["cpu_baseline"]["spread_O3_native"][0]) / 1e3,
       max(hi, d2["cpu_baseline"]["spread_O2"][1], d2["cpu_baseline"]["spread_O3_native"][1]) / 1e3, d["value"] / cb["value_1core"])) + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- frac:begin -->).*?(<!-- frac:end -->)", lambda m: m.group(1) + "%.2g of 8 TB/s" % rf["frac"] + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- target:begin -->).*?(<!-- target:end -->)", lambda m: m.group(1) + "%.1f ×" % (d["value"] / 1e5) + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- lat:begin -->).*?(<!-- lat:end -->)", lambda m: m.group(1) + (
    "with the selected trajectory on the host too (SURVEY §8(d)'s tick) %.1f µs median, %.1f µs p99; set-point alone %.1f µs median, %.1f µs p99 "
    "— both on the library's own clock (`pmaf_get_winner_path_times_us`, `pmaf_get_tick_times_us`)" % (wp.get("median", float("nan")), wp.get("p99", float("nan")), lib["median"], lib["p99"])) + m.group(2), r, flags=re.S)
open(rp, "w").write(r)
print(block)

"""Regenerates the measured-numbers block of DESIGN.md (between the `numbers:begin` / `numbers:end` markers), the BOUND
paragraphs of its section 4 (`bound_w64` / `bound_mw` / `bound_grp` markers: instructions per wave-step by class, issue slots,
issue fraction, traffic ratio, FP64 fraction -- from the PMC summaries), the scaling table of section 6 (`scaling` markers,
from <round>_scaling_emulated.json) and the generated blocks of README.md from the committed evidence of the round
(ROUND in the environment, default r6): profiles/<round>_bench_c2_driver_flags.json (the driver's command),
_bench_c2.json, _bench_c3.json, _bench_c5x8.json, _c{2,3,5}_trace.txt (rocprofv3 --kernel-trace --stats), _c{2,3,5}_pmc{1,2}.txt,
traffic.json, _regime.json, _ticklat.txt, _cpu_bench_c2_run{1,2}.json.
FAILS when the bench line's roofline.traffic is not the value of profiles/traffic.json (a bench line taken before the PMC
passes were regenerated must not be quoted beside them: VERDICT r4 weak 5). usage: python tools/fill_numbers.py"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
R = os.environ.get("ROUND", "r6")


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
d2 = load("%s_bench_c2.json" % R)
c3, c5 = load("%s_bench_c3.json" % R), load("%s_bench_c5x8.json" % R)
tj = json.load(open(os.path.join(P, "traffic.json")))
cf = d["configs"]
rf, fv, cb = d["roofline"], d["fp64_valu"], d["cpu_baseline"]
# ---- consistency gate: the line's traffic must be traffic.json's
key = "C2:%s" % rf["kernel"]
if key not in tj or rf["traffic"] is None or abs(rf["traffic"] - tj[key]["traffic_bytes_per_launch"]) > 0.5:
    sys.exit("fill_numbers: roofline.traffic of profiles/%s_bench_c2_driver_flags.json (%s) is not profiles/traffic.json's %s (%s): "
             "re-run the bench line after tools/%s_collect.sh prof" % (R, rf["traffic"], key, tj.get(key, {}).get("traffic_bytes_per_launch"), R))
t2, n2 = trace_avg("c2", "k_rollout_w64<1, 2, true, true>")
t3, n3 = trace_avg("c3", "k_rollout_mw<2, 2, true, false>")
t5, n5 = trace_avg("c5", "k_rollout_grp<16, 2, 2>")
sp = d["setpoint_latency_us"]
wp = d.get("tick_with_winner_path_us") or {}
rows = [
    ("**C2 64 × 200 × 32** (headline, `--steps 20 --warmup 5`)", d["value"], d["ms_per_step"], "`k_rollout_w64<1,2,true,true>` %.1f / %.1f (%d calls)" % (rf["avg_kernel_us"], t2, n2)),
    ("C1 16 × 100 × 9", cf["C1"]["rollouts_per_s"], cf["C1"]["ms_per_tick"], "`k_rollout_w64<1,2,true,true>` %.1f" % cf["C1"]["avg_kernel_us"]),
    ("C3 256 × 500 × 128 (sub-record / own run)", cf["C3"]["rollouts_per_s"], cf["C3"]["ms_per_tick"], "`k_rollout_mw<2,2,true,false>` %.1f / %.1f / %.1f (%d calls)" % (cf["C3"]["avg_kernel_us"], c3["roofline"]["avg_kernel_us"], t3, n3)),
    ("C5 8 × 1024 × 200 × 32 on one GPU (sub-record / own run)", cf["C5_sharded"]["rollouts_per_s"], cf["C5_sharded"]["ms_per_tick"], "`k_rollout_grp<16,2,2>` %.1f / %.1f / %.1f (%d calls)" % (cf["C5_sharded"]["avg_kernel_us"], c5["roofline"]["avg_kernel_us"], t5, n5)),
    ("C4 dual arm 2 × 256 × 200 × 32, one GPU, peer mailboxes", cf["C4"]["rollouts_per_s"], cf["C4"]["ms_per_tick"], "`k_rollout_w64<1,2,true,true>` %.1f; header wait %.2f µs median / %.2f p99, publish %.2f µs" % (
        cf["C4"]["avg_kernel_us"], cf["C4"]["header_exchange_us"]["wait_median"], cf["C4"]["header_exchange_us"]["wait_p99"], cf["C4"]["header_exchange_us"]["publish_median"])),
]
try:   # BASELINE C5's per-GPU load at 4 GPUs: two scenes in one handle = two wave-per-agent rollouts per SIMD
    c52 = load("%s_bench_c5x2.json" % R)
    t52, n52 = trace_avg("c5x2", "k_rollout_w64_sliced<2>")
    rows.append(("C5, two scenes per GPU: 2 × 1024 × 200 × 32 (own run; priority slices)", c52["value"], c52["ms_per_step"],
                 "`k_rollout_w64_sliced<2>` %.1f / %.1f (%d calls)" % (c52["roofline"]["avg_kernel_us"], t52, n52)))
except (OSError, ValueError, KeyError):
    pass
ts = cf.get("task_static1") or {}
if "rollouts_per_s" in ts:
    rows.append(("`task_static1`: 10 agents × ≤ 1499 steps × 9 (the shipped task; h_eff %.0f)" % ts["h_eff"], ts["rollouts_per_s"], ts["ms_per_tick"],
                 "`k_rollout_w64<1,2,true,true>` %.0f = %.1f %% of the 10 ms period" % (ts["avg_kernel_us"], 100 * ts["regime"]["share_of_the_control_period"])))
for name, kk_, kn in (("C2 contracted (opt-in; tolerance parity met)", "C2_contracted", "k_rollout_w64<1,3,true,true>"),
                      ("C3 contracted (tolerance parity met)", "C3_contracted", "k_rollout_mw<2,3,true,false>"),
                      ("C5 × 8 contracted (`parity_met: false`: 3.2e-3 m on scene 1)", "C5_sharded_contracted", "k_rollout_grp<16,2,3>")):
    if kk_ in cf and "rollouts_per_s" in cf[kk_]:
        rows.append((name, cf[kk_]["rollouts_per_s"], cf[kk_]["ms_per_tick"], "`%s` %.1f" % (kn, cf[kk_]["avg_kernel_us"])))
out = []
kt = rf.get("kernel_timing") or {}
out.append("One MI355X, round 6 (the kernels are round 5's; ABI 7). `profiles/%s_bench_c2_driver_flags.json` = the driver's command "
           "(`python bench.py --steps 20 --warmup 5`: %d blocks, %.2f s timed; HIP events on every %s-th rollout launch: %s of %s) with its "
           "sub-records; rocprofv3 `--kernel-trace --stats` of the same workloads: `profiles/%s_c{2,3,5}_trace.txt`; PMC passes "
           "`profiles/%s_c*_pmc*.txt` → `profiles/traffic.json`.\n"
           % (R, d["timing"]["blocks"], d["timing"]["timed_s"], kt.get("every", 8), kt.get("launches_timed_with_hip_events", "?"),
              kt.get("launches_in_timed_region", "?"), R, R))
out.append("| config | rollouts/s | ms/tick | rollout kernel µs per launch (HIP events in the bench [/ own run] / rocprofv3 avg) |\n|---|---|---|---|")
for name, v, ms, kk_ in rows:
    out.append("| %s | %s | %.4f | %s |" % (name, k(v), ms, kk_))
out.append("")
lib = sp.get("in_library")
lat_extra = ""
try:
    tl = open(os.path.join(P, "%s_ticklat.txt" % R)).read()
    m_open = re.search(r"open loop\s+\| around the calls median ([0-9.]+) p99 ([0-9.]+) us \| pmaf_tick alone \(library clock\) set-point median ([0-9.]+)", tl)
    m_closed = re.search(r"closed loop\s+\| around the calls median ([0-9.]+) p99 ([0-9.]+) us \| pmaf_tick alone \(library clock\) set-point median ([0-9.]+)", tl)
    m_plain = re.search(r"events False obstacles False winner path False \| enqueue median [0-9.]+ p99 [0-9.]+ \| set-point median ([0-9.]+) p99 ([0-9.]+)", tl)
    if m_open and m_closed:
        lat_extra = (" **Closed loop** (`pmaf_set_real_position` leaves the measured position in pinned memory, the manager kernel reads it: "
                     "no stream sync, no copy command): `pmaf_tick` on the library clock %s µs median against %s open loop "
                     "(`profiles/%s_ticklat.txt`)." % (m_closed.group(3), m_open.group(3), R))
    if m_plain:
        lat_extra += " Without event timing of the launches the set-point takes %s µs median / %s p99." % (m_plain.group(1), m_plain.group(2))
except OSError:
    pass
try:   # the same question at the C++ boundary, no interpreter (tests/cpp/facade_tick lat)
    fl = open(os.path.join(P, "%s_facade_latency.txt" % R)).read()
    sec = fl.split("## agents, max_prediction_steps: 64 201")[1]
    g_ = lambda what: re.search(re.escape(what) + r"\s+median\s+([0-9.]+)\s+p90\s+[0-9.]+\s+p99\s+([0-9.]+)", sec).groups()
    o_, c_, f_ = g_("planTick, open loop"), g_("setRealEEAgentPosition + planTick, closed loop"), g_("the node's five calls (stop ... start)")
    lat_extra += (" **At the C++ boundary** (`tests/cpp/facade_tick lat`, one `planCallback` through the facade, 64 agents, 1000 samples, "
                  "`profiles/%s_facade_latency.txt`): `planTick` %s µs median / %s p99 open loop, **%s / %s closed loop** (`setRealEEAgentPosition` "
                  "costs nothing extra since ABI 6; it was a stream sync + copy), the node's five individual calls %s / %s µs (three manager launches, each awaited through the mailbox, inputs by value / pinned memory; 106 µs in round 4)."
                  % (R, o_[0], o_[1], c_[0], c_[1], f_[0], f_[1]))
except (OSError, IndexError, AttributeError):
    pass
out.append("**Tick latency** on an idle stream, library clock. SURVEY §8(d)'s tick — host call → best index, set-point AND the selected "
           "agent's scored path on the host (`pmaf_enable_winner_path`, %d B at C2), %d ticks — **median %.1f µs, p90 %.1f, p99 %.1f, max %.1f**. "
           "Set-point alone (%d samples): median %.1f µs, p90 %.1f, p99 %.1f, max %.1f (%.1f µs of it = handing both launches to the stream); "
           "back-to-back tick median %.1f µs.%s"
           % (wp.get("path_bytes", 0), wp.get("n", 0), wp.get("median", float("nan")), wp.get("p90", float("nan")), wp.get("p99", float("nan")),
              wp.get("max", float("nan")), sp.get("n", 0), lib["median"], lib["p90"], lib["p99"], lib["max"], lib["enqueue_median"],
              d["tick_latency_us"]["median"], lat_extra))
try:
    dy = load("%s_bench_c2_dynamic.json" % R)
    out.append("")
    out.append("Inputs are resident in HBM when the timed region starts. With MOVING obstacles the caller hands a new list over on every tick "
               "(%d B, read by the manager kernel out of mapped pinned memory): %s rollouts/s, %.4f ms/tick "
               "(`profiles/%s_bench_c2_dynamic.json`) — the PCIe-inclusive rate of this path."
               % (7 * 8 * (d["config"]["obstacles"] + 1), k(dy["value"]), dy["ms_per_step"], R))
except OSError:
    pass
out.append("")
out.append("**Roofline of the dominant kernel (C2 launch).** Algorithmic bytes (SURVEY §8d) %d B ÷ %.1f µs = %.3f GB/s = **%.3g of 8 TB/s** "
           "(`roofline.frac`; rocprofv3 average of the same kernel %.1f µs). FP64-VALU: %d measured FP64 operations per agent-step "
           "(`oracle/flopcount.cpp`, %.1f %% of the agent-steps with an obstacle inside the shell) → %.3f TFLOP/s = %.3f %% of 78.6 TF. HBM traffic "
           "from the PMC passes (FETCH_SIZE × 2 + WRITE_SIZE, calibrated): %.2f MB per launch = **%.2f × algorithmic** (the cost pass re-reads the "
           "path; four orders below any limit). The north star's \"≥ 40 %% HBM\" is structurally unattainable for this algorithm (SURVEY §8d); "
           "its throughput target (100 k rollouts/s at C2) is met %.2f ×."
           % (rf["algorithmic_bytes_per_launch"], rf["avg_kernel_us"], rf["achieved"], rf["frac"], t2, round(fv["flops_per_agent_step"]),
              100.0 * (fv.get("in_shell_step_fraction") or 0.0), fv["achieved_tflops"], 100.0 * fv["frac"], rf["traffic"] / 1e6,
              rf["traffic"] / rf["algorithmic_bytes_per_launch"], d["value"] / 1e5))
out.append("")
# ---- CPU baseline
runs = []
for nme in ("%s_cpu_bench_c2_run1.json" % R, "%s_cpu_bench_c2_run2.json" % R):
    try:
        r_ = load(nme)
        runs.append(max(b["multi"]["best"] for b in r_["builds"].values() if "multi" in b))
    except (OSError, ValueError):
        pass
bests = [cb["best"], d2["cpu_baseline"]["best"]] + runs
out.append("**CPU baseline** (`cpu_baseline`, kind `port`: `oracle/cpu_bench.py`, the oracle in a process of its own on the box's host: %s, "
           "%d logical CPUs, %s in the affinity mask, cgroup CPU quota %s → **%d physical cores used, one pinned OpenMP thread each** (`GOMP_CPU_AFFINITY`), never "
           "more threads than agents; ≥ 3 s warm-up; `-O2` and `-O3 -march=native`, bit-identical results). At C2: best repetition **%s**, median %s "
           "rollouts/s on %d threads, %.0f %% of the repetitions within 10 %% of the best; one core: best %s. Reproducibility of `best` on this box: "
           "%s rollouts/s over %d runs (the two bench lines and two stand-alone runs back to back; spread %.0f %%). The driver's earlier records of "
           "the un-pinned measurement: 254 k (r03, 64 threads) and 60 k (r04, 32 threads, repetitions from 50 k to 177 k). **GPU / CPU at C2 = "
           "%.2f × the port's best, %.2f × its median, %.0f × one core**; C3 %.0f ×, C5 × 8 %.0f × (best). C2 is 64 independent 200-step chains — the "
           "shape where a GPU has the least to offer; the claim here is parity and an issue-bound step, not the ratio."
           % (cb["cpu_model"], cb["host_cpus"], cb["affinity_cpus"], cb["cgroup_cpu_quota"], cb["physical_cores_allowed"], k(cb["best"]), k(cb["value"]), cb["cores"],
              100 * cb["share_of_repetitions_within_10pct_of_best"], k(cb["best_1core"]),
              " / ".join(k(b) for b in bests), len(bests), 100.0 * (max(bests) - min(bests)) / max(bests),
              d["value"] / cb["best"], d["value"] / cb["value"], d["value"] / cb["best_1core"],
              c3["value"] / c3["cpu_baseline"]["best"], c5["value"] / c5["cpu_baseline"]["best"]))
out.append("")
# ---- regime
try:
    rg = json.load(open(os.path.join(P, "%s_regime.json" % R)))
    out.append("**The regime** (`tools/regime.py` → `profiles/%s_regime.json`; full-horizon rollouts, H = %d, %s): a rollout is ONE dependent chain; "
               "more lanes do not shorten it. Per step of a chain the MI355X wave is SLOWER than one x86 core up to a few dozen obstacles (at 128 the two-wave split kernel draws level); the GPU wins by running thousands of chains at once." % (R, rg["horizon"], rg["cpu_model"]))
    out.append("")
    out.append("| field obstacles M | one CPU core, ns per agent-step | GPU, ns per step of a chain (64 agents) | GPU ÷ CPU per chain | agents from which one MI355X beats a 64-core host (one agent per core) | aggregate agent-steps/µs at 8 192 agents: GPU vs 64 cores |\n|---|---|---|---|---|---|")
    for r_ in rg["rows"]:
        cx = r_["agents_at_which_the_gpu_overtakes_a_host_of_C_cores"]
        out.append("| %d | %.0f | %.0f | %.1f × | %s | %.0f vs %.0f |" % (
            r_["field_obstacles"], r_["cpu_ns_per_agent_step_one_core"], r_["gpu_ns_per_step_of_a_chain"], r_["gpu_chain_over_cpu_chain"],
            cx.get("64"), r_["gpu_agent_steps_per_us_at_8192"], r_["cpu_agent_steps_per_us_64_cores"]))
    if "cpu_port" in ts and "best" in ts["cpu_port"]:
        cp = ts["cpu_port"]
        out.append("")
        out.append("At the reference's OWN operating point (`configs.task_static1`: 10 agents, `max_prediction_steps` 1500, 9 + 1 obstacles, 100 Hz; "
                   "`B/config/tasks/dual_arms_static1.yaml:2,15,19`) the rollout launch takes **%.0f µs** (h_eff %.0f; %.1f %% of the 10 ms period; "
                   "%.2f µs per step of the longest chain) against **%.0f µs per tick for the CPU port with one thread per agent** (%d threads, best "
                   "repetition) — the GPU path is **%.1f × slower per rollout** there, and both fit the control period several times over. What the GPU "
                   "path buys at that size is not speed: the set-point latency above does not depend on the horizon, the rollouts always run to "
                   "their guard (the reference's are cut by wall clock), and the ten host cores the reference's threads spin on are free."
                   % (ts["avg_kernel_us"], ts["h_eff"], 100 * ts["regime"]["share_of_the_control_period"], ts["regime"]["us_per_step_of_the_longest_chain"],
                      cp["tick_us_best"], cp["cores"], 1.0 / cp["gpu_over_cpu_best"]))
except (OSError, KeyError) as e:
    out.append("(regime table unavailable: %s)" % e)
block = "\n".join(out) + "\n"


# ---- section 4: what bounds each tuned kernel, from the PMC passes (no hand-carried number)
def pmc(tag, i, kernel_prefix):
    """counter -> average per dispatch of the kernel whose name starts with kernel_prefix (tools/prof_summary.py's table)"""
    vals = {}
    for line in open(os.path.join(P, "%s_%s_pmc%d.txt" % (R, tag, i))):
        f = [x.strip() for x in line.split("|")]
        if len(f) == 5 and kernel_prefix in f[0] and f[1] not in ("counter", "calls"):
            try:
                vals[f[1]] = float(f[3])
            except ValueError:
                pass
    return vals


def bound(tag, kernel_prefix, line, tj_key, agents_per_wave):
    a, b = pmc(tag, 1, kernel_prefix), pmc(tag, 2, kernel_prefix)
    if not a or "SQ_WAVES" not in a:
        sys.exit("fill_numbers: no PMC summary for %s in profiles/%s_%s_pmc1.txt" % (kernel_prefix, R, tag))
    waves, steps = a["SQ_WAVES"], line["h_eff"]
    per = lambda c: a.get(c, 0.0) / waves / steps
    valu, salu, lds = per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS")
    vmem = per("SQ_INSTS_VMEM_RD") + per("SQ_INSTS_VMEM_WR")
    slots = per("SQ_WAVE_CYCLES")
    tot = valu + salu + lds + vmem
    t = tj[tj_key]
    # consistency gate (like the traffic one): the PMC passes and the bench line must describe the same kernel launch
    rfl = line["roofline"]
    if abs(t["waves_per_launch"] - waves) > 1.0 or kernel_prefix.replace(" ", "") not in rfl["kernel"].replace(" ", ""):
        sys.exit("fill_numbers: %s: PMC waves %.1f vs traffic.json %.1f / bench kernel %s" % (tag, waves, t["waves_per_launch"], rfl["kernel"]))
    fv_ = line["fp64_valu"]
    return dict(waves=waves, steps=steps, valu=valu, salu=salu, lds=lds, vmem=vmem, tot=tot, slots=slots, issue=tot / slots,
                vaf=t["valu_active_frac_of_wave_cycles"], ratio=t["traffic_bytes_per_launch"] / rfl["algorithmic_bytes_per_launch"],
                hbm=rfl["frac"], tf=fv_["achieved_tflops"], fpfrac=fv_["frac"], us=rfl["avg_kernel_us"],
                flop_per_lane_op=fv_["flops_per_agent_step"] * agents_per_wave / (valu * 64.0))


B2 = bound("c2", "k_rollout_w64<1, 2, true, true>", d, "C2:k_rollout_w64<1, 2, true, true>", 1)
B3 = bound("c3", "k_rollout_mw<2, 2, true, false>", c3, "C3:k_rollout_mw<2, 2, true, false>", 0.5)
B5 = bound("c5", "k_rollout_grp<16, 2, 2>", c5, "C5x8:k_rollout_grp<16, 2, 2>", 4)
bw64 = ("**Bound** (C2; generated from `profiles/%s_c2_pmc1.txt`, `_pmc2.txt`, `traffic.json`, `%s_bench_c2_driver_flags.json`): %d waves on 1 024 SIMDs; per "
        "wave-step %.0f VALU + %.0f SALU + %.0f LDS + %.1f VMEM = %.0f instructions in %.0f issue slots (`SQ_WAVE_CYCLES` ÷ waves ÷ steps) = "
        "**%.0f %% of the one-wave issue limit**; `SQ_ACTIVE_INST_VALU` / `SQ_WAVE_CYCLES` = %.2f. The launch (%.1f µs) is as long as its slowest "
        "agent + ≈ 10 µs. HBM traffic %.2f × algorithmic (the cost pass re-reads the path; the LDS-ring fix is 2–5 %% slower, NOTES §2): "
        "irrelevant at %.2g of peak; FP64 %.3f TF = %.2f %% of 78.6 TF."
        % (R, R, round(B2["waves"]), B2["valu"], B2["salu"], B2["lds"], B2["vmem"], B2["tot"], B2["slots"], 100 * B2["issue"], B2["vaf"], B2["us"],
           B2["ratio"], B2["hbm"], B2["tf"], 100 * B2["fpfrac"]))
bmw = ("**Bound** (C3; generated from `profiles/%s_c3_pmc1.txt`, `_pmc2.txt`, `traffic.json`, `%s_bench_c3.json`): the chain is as long as in the one-wave "
       "kernel (sweep → terms → ordered sum of ≈ 57 dependent accumulates → tail); only the per-obstacle ISSUE is spread over the block's waves: "
       "%d waves, per wave-step %.0f VALU + %.0f SALU + %.0f LDS + %.1f VMEM = %.0f instructions in %.0f issue slots = %.0f %% of a wave's issue "
       "limit (`SQ_ACTIVE_INST_VALU` / `SQ_WAVE_CYCLES` = %.2f); launch %.1f µs, %.2g of HBM peak, traffic %.2f × algorithmic, FP64 %.2f TF = %.2f %% "
       "of peak. History: C3 977 → 930 µs against the two-slot one-wave kernel, 129…256 obstacles −26…32 %% (`profiles/r4_ab_mw.txt`); barrier wait "
       "0.4–3 %% of the loop."
       % (R, R, round(B3["waves"]), B3["valu"], B3["salu"], B3["lds"], B3["vmem"], B3["tot"], B3["slots"], 100 * B3["issue"], B3["vaf"], B3["us"], B3["hbm"],
          B3["ratio"], B3["tf"], 100 * B3["fpfrac"]))
bgrp = ("**Bound** (C5 × 8; generated from `profiles/%s_c5_pmc1.txt`, `_pmc2.txt`, `traffic.json`, `%s_bench_c5x8.json`): FP64-VALU issue — %d waves = two "
        "per SIMD, per wave-step (4 agents) %.0f VALU + %.0f SALU + %.0f LDS + %.1f VMEM instructions; `SQ_ACTIVE_INST_VALU` / `SQ_WAVE_CYCLES` = %.2f per "
        "wave ⇒ the SIMD's VALU ≈ %.0f %% busy; launch %.1f µs, %.2f TF = **%.2f %% of FP64 peak**, %.2f flop per lane-operation (the per-agent part "
        "is replicated over the group's lanes; split / helper wave / LPA 8 were built, measured slower, rejected: NOTES §2); %.2g of HBM peak, "
        "traffic %.2f × algorithmic."
        % (R, R, round(B5["waves"]), B5["valu"], B5["salu"], B5["lds"], B5["vmem"], B5["vaf"], 200 * B5["vaf"], B5["us"], B5["tf"], 100 * B5["fpfrac"],
           B5["flop_per_lane_op"], B5["hbm"], B5["ratio"]))

# ---- section 6: the emulated strong-scaling curve of BASELINE C5
sc_block = "(no profiles/%s_scaling_emulated.json)" % R
try:
    se = json.load(open(os.path.join(P, "%s_scaling_emulated.json" % R)))
    rows6 = ["| GPUs | scenes per GPU | lanes per agent | rollout kernel µs (slowest rank) | ms per tick | aggregate rollouts/s | speed-up | efficiency vs 1 GPU |",
             "|---|---|---|---|---|---|---|---|"]
    for n in ("1", "2", "4", "8"):
        r_ = se["predicted_by_n"][n]
        rows6.append("| %s | %d | %d | %.1f | %.4f | %s | %.2f × | %.0f %% |" % (
            n, r_["populations_per_gpu"], r_["lanes_per_agent"], r_["kernel_us_slowest_rank"], r_["ms_per_tick"], k(r_["rollouts_per_s"]),
            r_["speedup_vs_1gpu"], 100 * r_["efficiency_vs_1gpu"]))
    ag = se.get("allgather", {})
    r1_ = ag.get("rccl_one_rank_free_cus") or {}
    r8_ = ag.get("rccl_one_rank_under_c5x8") or {}
    h2_ = ag.get("two_ranks_one_gpu_host_transport") or {}
    ser = se.get("if_the_allgather_were_serialised") or {}
    sc_block = ("**Predicted 1 → 8-GPU curve of BASELINE C5 (strong scaling), emulated on ONE GPU** (`bench.py`: `scaling_c5.prediction`; "
                "`profiles/%s_scaling_emulated.json`): every rank's share of the eight scenes run in a handle of its own, one after the other; the "
                "job's tick at N GPUs is its slowest rank's; the winner-record all-gather is not on the tick's critical path.\n\n" % R
                + "\n".join(rows6) + "\n\n"
                + "**Why it saturates at ≈ %.1f ×**: a rollout is one dependent 200-step chain; with one scene per GPU (1 024 waves, one per SIMD) the "
                  "launch sits on the chain floor of ≈ %.0f µs and cannot get shorter, while ONE GPU already runs all eight scenes in %.2f ms by "
                  "packing four agents into a wave at two waves per SIMD. Strong scaling of this configuration is bounded by construction, "
                  "not by communication: the curve to expect from `SCALE_rNN.json` is this table, and the line's `scaling_c5` block (not `value`, "
                  "which is BASELINE C2 replicated per GPU: weak scaling, ≈ N × by construction) is where to read it. "
                  % (se["predicted_by_n"]["8"]["speedup_vs_1gpu"], se["chain_floor_us"], se["predicted_by_n"]["1"]["ms_per_tick"])
                + ("The all-gather where it can be measured here: `ncclAllGather` with a ONE-rank RCCL communicator costs %.1f µs median / %.1f p99 on "
                   "the exchange stream when its kernel gets a CU at once (C2)" % (r1_["median_us"], r1_["p99_us"]) if r1_ else "")
                + ("; beside the 8-scene rollout it completes only after %.0f µs — the group kernel's 2 × 252 VGPRs fill every SIMD, the collective's "
                   "kernel waits for the first waves to retire — and the tick does not wait for it (%.4f ms per tick with it, %.4f without)"
                   % (r8_["median_us"], r8_["ms_per_tick_with_it"], r8_["ms_per_tick_without"]) if r8_ else "")
                + ("; two ranks sharing this GPU over the host transport: %.0f µs (plumbing, not xGMI)" % h2_["median_us"] if h2_ else "")
                + (". Were the collective serialised behind every tick at the free-CU cost per ring step, the 8-GPU efficiency would be %.0f %% instead "
                   "of %.0f %%." % (100 * ser["8"]["efficiency_vs_1gpu"], 100 * se["predicted_by_n"]["8"]["efficiency_vs_1gpu"]) if ser else ".")
                + " Weak scaling (more scenes than GPUs × 8, or BASELINE C2 per GPU) has no such bound: ranks share nothing.")
except (OSError, KeyError, ValueError) as e:
    sc_block = "(scaling table unavailable: %s)" % e

p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
s = re.sub(r"(<!-- numbers:begin[^\n]*-->\n).*?(<!-- numbers:end -->)", lambda m: m.group(1) + block + m.group(2), s, flags=re.S)
for name, text in (("bound_w64", bw64), ("bound_mw", bmw), ("bound_grp", bgrp), ("scaling", sc_block)):
    if ("<!-- %s:begin -->" % name) not in s:
        sys.exit("fill_numbers: DESIGN.md has no %s markers" % name)
    s = re.sub(r"(<!-- %s:begin -->\n).*?(<!-- %s:end -->)" % (name, name), lambda m: m.group(1) + text + "\n" + m.group(2), s, flags=re.S)
open(p, "w").write(s)

# README
rp = os.path.join(ROOT, "README.md")
r = open(rp).read()


def cpu_cell(c):
    b = c["cpu_baseline"]
    return "%.0f k best, %.0f k median (%d) / %.1f k" % (b["best"] / 1e3, b["value"] / 1e3, b["cores"], b["best_1core"] / 1e3)


tab = ["| BASELINE config | rollouts/s | tick | rollout kernel | CPU port: best repetition, median (pinned threads) / 1 core |", "|---|---|---|---|---|",
       "| C2: 64 agents × 200 steps × 32 obstacles | %.0f k | %.3f ms | %.0f µs | %s |" % (d["value"] / 1e3, d["ms_per_step"], rf["avg_kernel_us"], cpu_cell(d)),
       "| C3: 256 × 500 × 128 | %.0f k | %.2f ms | %.2f ms | %s |" % (c3["value"] / 1e3, c3["ms_per_step"], c3["roofline"]["avg_kernel_us"] / 1e3, cpu_cell(c3)),
       "| C5: 8 × 1024 × 200 × 32 (one GPU) | %.1f M | %.2f ms | %.2f ms | %s |" % (c5["value"] / 1e6, c5["ms_per_step"], c5["roofline"]["avg_kernel_us"] / 1e3, cpu_cell(c5))]
r = re.sub(r"(<!-- headline:begin -->\n).*?(<!-- headline:end -->)", lambda m: m.group(1) + "\n".join(tab) + "\n" + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- ratio:begin -->).*?(<!-- ratio:end -->)", lambda m: m.group(1) + (
    "%.2f × the best repetition (%.2f × the median) of the multi-threaded CPU port of the same algorithm on the box's %s with one pinned thread per "
    "allowed physical core -- %d of them: the cgroup's CPU quota (`best` reproduces within %.0f %% over %d runs on that box; the un-pinned measurement of rounds 3 / 4 ranged from 60 k to "
    "254 k rollouts/s between driver runs) — and %.0f × one core"
    % (d["value"] / cb["best"], d["value"] / cb["value"], cb["cpu_model"], cb["physical_cores_allowed"], 100.0 * (max(bests) - min(bests)) / max(bests), len(bests),
       d["value"] / cb["best_1core"])) + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- frac:begin -->).*?(<!-- frac:end -->)", lambda m: m.group(1) + "%.2g of 8 TB/s" % rf["frac"] + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- target:begin -->).*?(<!-- target:end -->)", lambda m: m.group(1) + "%.1f ×" % (d["value"] / 1e5) + m.group(2), r, flags=re.S)
r = re.sub(r"(<!-- lat:begin -->).*?(<!-- lat:end -->)", lambda m: m.group(1) + (
    "with the selected trajectory on the host too (SURVEY §8(d)'s tick) %.1f µs median, %.1f µs p99; set-point alone %.1f µs median, %.1f µs p99 "
    "— both on the library's own clock (`pmaf_get_winner_path_times_us`, `pmaf_get_tick_times_us`)" % (wp.get("median", float("nan")), wp.get("p99", float("nan")), lib["median"], lib["p99"])) + m.group(2), r, flags=re.S)
try:
    se_ = json.load(open(os.path.join(P, "%s_scaling_emulated.json" % R)))["predicted_by_n"]
    r = re.sub(r"(<!-- scale:begin -->).*?(<!-- scale:end -->)", lambda m: m.group(1) + " / ".join(
        "%s GPU%s %.2f ms per tick (%.0f %%)" % (n, "" if n == "1" else "s", se_[n]["ms_per_tick"], 100 * se_[n]["efficiency_vs_1gpu"]) for n in ("1", "2", "4", "8"))
        + ", i.e. %.1f × at 8 GPUs" % se_["8"]["speedup_vs_1gpu"] + m.group(2), r, flags=re.S)
except (OSError, KeyError, ValueError):
    pass
open(rp, "w").write(r)
print(block)

#!/bin/bash
# Round-4 evidence run on the GPU box (gpurun -- bash tools/r4_evidence.sh [part ...]): bench lines, rocprofv3 trace + PMC
# summaries (C2 / C3 / C5 x 8, strict; C2 / C5 x 8 contracted), per-policy timing on one box, tick-latency table,
# tolerance report, sanitizer passes, fuzz campaign, the GPU suite. Everything lands in gpurun_out/r4/; what is kept is
# copied to profiles/r4_* by tools/r4_collect.sh.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4
mkdir -p $O
PARTS=${@:-bench prof misc asan tests}
for part in $PARTS; do case $part in
bench)
  python bench.py --steps 20 --warmup 5 > $O/bench_c2_driver_flags.json 2> $O/bench.err
  python bench.py > $O/bench_c2.json 2>> $O/bench.err
  python bench.py --config C3 --steps 400 --only-headline --cpu-seconds 6 > $O/bench_c3.json 2>> $O/bench.err
  python bench.py --config C5 --populations 8 --steps 400 --only-headline --cpu-seconds 6 --flop-ticks 2 > $O/bench_c5x8.json 2>> $O/bench.err
  python bench.py --dynamic --only-headline --cpu-seconds 0 > $O/bench_c2_dynamic.json 2>> $O/bench.err
  python bench.py --time-every 1 --only-headline --cpu-seconds 0 --flop-ticks 0 > $O/bench_c2_every_launch_timed.json 2>> $O/bench.err
  PMAF_BENCH_BACKEND=gloo PMAF_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --cpu-seconds 0 --flop-ticks 0 2>> $O/bench.err | grep "^{" | tail -1 > $O/bench_2ranks_1gpu_selfspawn.json
  PMAF_BENCH_FORCE_DIST=1 MASTER_PORT=29531 python bench.py --only-headline --cpu-seconds 0 --flop-ticks 0 2>> $O/bench.err | grep "^{" | tail -1 > $O/bench_c2_rccl_1rank.json
  ;;
prof)
  bash tools/gpu_prof.sh r4_c2 > /dev/null 2>&1
  bash tools/gpu_prof.sh r4_c3 --config C3 --steps 400 > /dev/null 2>&1
  bash tools/gpu_prof.sh r4_c5 --config C5 --populations 8 --steps 400 > /dev/null 2>&1
  ;;
misc)
  python tools/policytime.py C1 C2 C3 C4 C5 --rounds 3 --out $O/policytime.json > $O/policytime.log 2>&1
  python tools/ticklat.py C2 600 > $O/ticklat.txt 2>&1
  { echo "# per-agent / per-wave rollout durations from the device clock, and launch duration against the slowest wave (final kernels of round 4)";
    python tools/agenttime.py C1 C2 C3; python tools/c5agents.py; python tools/launchgap.py; } 2>&1 | grep -v "^$\|amdgpu.ids" > $O/agent_times.txt
  bash tools/fuzz_campaign.sh > $O/fuzz_campaign.txt 2>&1
  ;;
soak)   # stability of the C2 tick loop: 240 s timed, block times
  python bench.py --only-headline --min-seconds 240 --steps 2000 --cpu-seconds 0 --flop-ticks 0 > $O/bench_c2_soak_240s.json 2>> $O/bench.err
  # (the split kernel: one s_barrier per step and wave -- 120 s of C3 ticks)
  python bench.py --config C3 --only-headline --min-seconds 120 --steps 500 --cpu-seconds 0 --flop-ticks 0 > $O/bench_c3_soak_120s.json 2>> $O/bench.err
  ;;
asan)   # (build first, here: bash tools/asan.sh build -- lib_asan/ and lib_bounds/ travel with the snapshot)
  bash tools/asan.sh run > /dev/null 2>&1; cp gpurun_out/r4_asan.txt $O/asan.txt
  ;;
tests)
  export PMAF_TOL_REPORT=$PWD/$O/tolerance_report.jsonl
  rm -f $PMAF_TOL_REPORT
  python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $O/gpu_tests_full.log
  tail -6 $O/gpu_tests_full.log > $O/gpu_tests.log
  ;;
esac; done
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    sp = (d.get("setpoint_latency_us") or {}).get("in_library") or {}
    wp = d.get("tick_with_winner_path_us") or {}
    print("value %.0f %s  n_gpus %d  ms/step %.4f  kernel %s %.1f us  h_eff %.1f  setpoint %s / p99 %s us  with path %s / %s  cpu %s" % (
        d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_kernel_us"], d["h_eff"],
        sp.get("median"), sp.get("p99"), wp.get("median"), wp.get("p99"), (d.get("cpu_baseline") or {}).get("value")))
    for k, c in (d.get("configs") or {}).items():
        print("   %-22s %10.0f rollouts/s  %.4f ms/tick  %s %.1f us  h_eff %.1f  hdr %s" % (
            k, c.get("rollouts_per_s", 0), c.get("ms_per_tick", 0), c.get("kernel"), c.get("avg_kernel_us", 0), c.get("h_eff", 0),
            (c.get("header_exchange_us") or {}).get("wait_median")))
except Exception as e:
    print("unreadable", e)
PY
done
cat $O/gpu_tests.log 2>/dev/null; tail -3 $O/bench.err 2>/dev/null

#!/bin/bash
# The round's evidence run on the GPU box (gpurun -- bash tools/evidence.sh [part ...]; ROUND=r6 by default): bench lines,
# the emulated C5 strong-scaling curve, rocprofv3 trace + PMC summaries (C2 / C3 / C5 x 8), the evaluation-order variant
# (full GPU suite + kernel times of both libraries), the regime table, tick latency open / closed loop, fuzz campaign, the
# GPU suite. Everything lands in gpurun_out/$ROUND/; what is kept is copied to profiles/${ROUND}_* by tools/collect.sh.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
R=${ROUND:-r6}
O=gpurun_out/$R
mkdir -p $O
PARTS=${@:-bench scaling prof misc variant tests}
for part in $PARTS; do case $part in
bench)
  python bench.py --steps 20 --warmup 5 > $O/bench_c2_driver_flags.json 2> $O/bench.err
  python bench.py > $O/bench_c2.json 2>> $O/bench.err
  python bench.py --config C3 --steps 400 --only-headline --cpu-seconds 6 > $O/bench_c3.json 2>> $O/bench.err
  python bench.py --config C5 --populations 8 --steps 400 --only-headline --cpu-seconds 6 --flop-ticks 2 > $O/bench_c5x8.json 2>> $O/bench.err
  python bench.py --dynamic --only-headline --cpu-seconds 0 > $O/bench_c2_dynamic.json 2>> $O/bench.err
  python bench.py --config C5 --populations 2 --steps 400 --only-headline --cpu-seconds 0 --flop-ticks 2 > $O/bench_c5x2.json 2>> $O/bench.err
  PMAF_BENCH_BACKEND=gloo PMAF_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --cpu-seconds 0 --flop-ticks 0 2>> $O/bench.err | grep "^{" | tail -1 > $O/bench_2ranks_1gpu_selfspawn.json
  PMAF_BENCH_BACKEND=gloo PMAF_BENCH_SINGLE_DEVICE=1 PMAF_BENCH_C4_HOST_COUPLED=1 python bench.py --gpus 2 --config C4 --cpu-seconds 0 --flop-ticks 0 2>> $O/bench.err | grep "^{" | tail -1 > $O/bench_c4_2ranks_host_coupled.json
  PMAF_BENCH_FORCE_DIST=1 MASTER_PORT=29531 python bench.py --only-headline --cpu-seconds 0 --flop-ticks 0 2>> $O/bench.err | grep "^{" | tail -1 > $O/bench_c2_rccl_1rank.json
  # the CPU baseline twice more, back to back: does `best` reproduce on this box?
  python oracle/cpu_bench.py --config C2 --budget 12 > $O/cpu_bench_c2_run1.json 2>> $O/bench.err
  python oracle/cpu_bench.py --config C2 --budget 12 > $O/cpu_bench_c2_run2.json 2>> $O/bench.err
  ;;
scaling)
  # BASELINE C5's strong-scaling curve emulated on this one GPU (bench.py: emulate_c5_scaling) + what the all-gather of
  # the winner records costs where it can be measured here: RCCL with ONE rank (the enqueue + kernel of ncclAllGather on
  # the exchange stream) and two ranks sharing this GPU over the host transport -> profiles/${R}_scaling_emulated.json
  PMAF_BENCH_FORCE_DIST=1 MASTER_PORT=29533 python bench.py --config C5 --shard --only-headline --cpu-seconds 0 --flop-ticks 0 --steps 200 2>> $O/bench.err | grep "^{" | tail -1 > $O/bench_c5_rccl_1rank.json
  PMAF_BENCH_BACKEND=gloo PMAF_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --config C5 --shard --only-headline --cpu-seconds 0 --flop-ticks 0 --steps 200 2>> $O/bench.err | grep "^{" | tail -1 > $O/bench_c5shard_2ranks_1gpu_host.json
  python tools/scaling_emulated.py $O/bench_c2_driver_flags.json $O/bench_c5_rccl_1rank.json $O/bench_c5shard_2ranks_1gpu_host.json $O/bench_c2_rccl_1rank.json > $O/scaling_emulated.json 2>> $O/bench.err
  ;;
prof)
  bash tools/gpu_prof.sh ${R}_c2 > /dev/null 2>&1
  bash tools/gpu_prof.sh ${R}_c3 --config C3 --steps 400 > /dev/null 2>&1
  bash tools/gpu_prof.sh ${R}_c5 --config C5 --populations 8 --steps 400 > /dev/null 2>&1
  bash tools/gpu_prof.sh ${R}_c5x2 --config C5 --populations 2 --steps 400 > /dev/null 2>&1   # two scenes per GPU: k_rollout_w64_sliced
  ;;
misc)
  python tools/regime.py --out $O/regime.json > $O/regime.txt 2>&1
  python tools/ticklat.py C2 600 > $O/ticklat.txt 2>&1
  # the same question at the C++ boundary (no interpreter): planTick open / closed loop, the node's five calls
  { echo "# tests/cpp/facade_tick lat: one planCallback through the C++ facade on an idle stream, static1 scene";
    for nc in "10 1500" "16 101" "64 201"; do echo "## agents, max_prediction_steps: $nc"; tests/cpp/facade_tick lat $nc 1000; done; } > $O/facade_latency.txt 2>&1
  { echo "# per-agent / per-wave rollout durations from the device clock (glibc-compatible exp since round 5)";
    python tools/agenttime.py C1 C2 C3; } 2>&1 | grep -v "^$\|amdgpu.ids" > $O/agent_times.txt
  bash tools/fuzz_campaign.sh > $O/fuzz_campaign.txt 2>&1
  ;;
variant)   # the right-associated library + oracle: the FULL 0-tolerance suite, and the kernel times of both libraries on one box
  PMAF_VARIANT=rassoc python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_build_variants.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $O/gpu_tests_rassoc.log
  for v in "" rassoc; do for c in C1 C2 C3 C4; do
    PMAF_VARIANT=$v python bench.py --config $c --only-headline --cpu-seconds 0 --flop-ticks 0 --min-seconds 0.5 --steps 200 2>> $O/bench.err | grep "^{" | tail -1 |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-7s %-3s %s %.1f us per launch, %.0f rollouts/s' % ('$v' or 'default', '$c', d['roofline']['kernel'], d['roofline']['avg_kernel_us'], d['value']))"
  done; PMAF_VARIANT=$v python bench.py --config C5 --populations 8 --only-headline --cpu-seconds 0 --flop-ticks 0 --min-seconds 0.5 --steps 200 2>> $O/bench.err | grep "^{" | tail -1 |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-7s %-3s %s %.1f us per launch, %.0f rollouts/s' % ('$v' or 'default', 'C5x8', d['roofline']['kernel'], d['roofline']['avg_kernel_us'], d['value']))"
  done > $O/variant_kernel_times.txt 2>&1
  ;;
tests)
  export PMAF_TOL_REPORT=$PWD/$O/tolerance_report.jsonl
  rm -f $PMAF_TOL_REPORT
  python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $O/gpu_tests_full.log
  tail -6 $O/gpu_tests_full.log > $O/gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
  ;;
esac; done
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    sp = (d.get("setpoint_latency_us") or {}).get("in_library") or {}
    cb = d.get("cpu_baseline") or {}
    print("value %.0f %s  n_gpus %d  ms/step %.4f  kernel %s %.1f us  h_eff %.1f  setpoint %s / p99 %s us  cpu median %s best %s (%s threads, share %s)" % (
        d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_kernel_us"], d["h_eff"],
        sp.get("median"), sp.get("p99"), cb.get("value"), cb.get("best"), cb.get("cores"), cb.get("share_of_repetitions_within_10pct_of_best")))
    for k, c in (d.get("configs") or {}).items():
        print("   %-22s %10.0f rollouts/s  %.4f ms/tick  %s %.1f us  h_eff %.1f  parity_met %s %s" % (
            k, c.get("rollouts_per_s", 0), c.get("ms_per_tick", 0), c.get("kernel"), c.get("avg_kernel_us", 0), c.get("h_eff", 0),
            c.get("parity_met"), ("cpu_port " + json.dumps(c["cpu_port"])[:200]) if "cpu_port" in c else ""))
except Exception as e:
    print("unreadable", e)
PY
done
cat $O/gpu_tests.log $O/gpu_tests_rassoc.log $O/variant_kernel_times.txt $O/regime.txt 2>/dev/null; tail -3 $O/bench.err 2>/dev/null

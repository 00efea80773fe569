#!/bin/bash
# round 3 (second session): core clock under the lone-wave load, perf-level experiment, PLAIN-step A/B, new parity test
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3b_clock.txt
{
echo "== rocm-smi clocks / perf level (idle)"
rocm-smi --showclocks --showperflevel 2>&1 | grep -v "^$" | head -30
hipcc --offload-arch=gfx950 -O2 tools/clockrate.hip -o /tmp/clockrate 2>&1 | tail -2
echo "== clockrate (auto)"; /tmp/clockrate; /tmp/clockrate
echo "== A/B PLAIN step (3 rounds, interleaved)"
for r in 1 2 3; do
  for ps in 1 0; do echo "-- PMAF_PLAIN_STEP=$ps"; PMAF_PLAIN_STEP=$ps python tools/quicktime.py C2:64 C3:64 C1:64 2>&1 | grep -v "^$"; done
done
echo "== clocks while C2 runs"
( python tools/quicktime.py C2:64 C2:64 C2:64 C2:64 > /tmp/qt.log 2>&1 & )
sleep 4; rocm-smi --showclocks 2>&1 | grep -i "sclk\|fclk\|mclk" | head; sleep 1; rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -3
sleep 6; cat /tmp/qt.log
echo "== setperflevel high"
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showclocks --showperflevel 2>&1 | grep -i "sclk\|perf" | head
/tmp/clockrate; /tmp/clockrate
python tools/quicktime.py C2:64 C3:64 C1:64 2>&1 | grep -v "^$"
echo "== setperfdeterminism 2400"
rocm-smi --setperfdeterminism 2400 2>&1 | tail -3
/tmp/clockrate
python tools/quicktime.py C2:64 C3:64 2>&1 | grep -v "^$"
echo "== back to auto"
rocm-smi --resetperfdeterminism 2>&1 | tail -2
rocm-smi --setperflevel auto 2>&1 | tail -2
/tmp/clockrate
python tools/quicktime.py C2:64 2>&1 | grep -v "^$"
} > $O 2>&1
python -m pytest tests/test_parity_gpu.py -x -q -k "general_step or c2_synthetic or c1_static or c3_256" > gpurun_out/r3b_newtest.log 2>&1
tail -3 gpurun_out/r3b_newtest.log
cat $O

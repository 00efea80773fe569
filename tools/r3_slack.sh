#!/bin/bash
# round-3 evidence run: thread trace availability, delay-insertion profiles of the C2 step loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/slack
( cd /tmp && timeout 120 rocprofv3 --att --kernel-trace -d /tmp/att_try -o a -- python $GRAFT_REPO_ROOT/tools/quicktime.py C2:64 ) > gpurun_out/slack/att_try.log 2>&1
echo "rc=$?" >> gpurun_out/slack/att_try.log
ls -R /tmp/att_try 2>/dev/null | head -20 >> gpurun_out/slack/att_try.log
python tools/slackprof.py gpurun_out/slack/c2_random.txt --type random > gpurun_out/slack/c2_random.log 2>&1
python tools/slackprof.py gpurun_out/slack/c2_goal.txt --type goal > gpurun_out/slack/c2_goal.log 2>&1
tail -5 gpurun_out/slack/att_try.log; head -40 gpurun_out/slack/c2_random.txt; tail -3 gpurun_out/slack/c2_random.log

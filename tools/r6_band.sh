#!/bin/bash
# round 6, first GPU call: (1) pin prerequisites probe (read-only), (2) pick_lpa band sweep 1025..2048 agents,
# (3) C5 with 8/4/2/1 populations per handle (per-GPU load at 1/2/4/8 GPUs)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6; mkdir -p $O
{
  echo "# probe for the pin's prerequisites on the GPU box (find, read-only)"
  echo "## Eigen/Dense:"; find / -xdev -type f -path '*Eigen/Dense' 2>/dev/null | head
  echo "## eigen3 dirs:"; find / -xdev -type d -name 'eigen3' 2>/dev/null | head
  echo "## dqrobotics/DQ.h:"; find / -xdev -path '*dqrobotics/DQ.h' 2>/dev/null | head
  echo "## libdqrobotics:"; find / -xdev -name 'libdqrobotics*' 2>/dev/null | head
  echo "## ros:"; ls /opt/ros 2>&1 | head -3
  echo "## /root/reference:"; ls /root/reference 2>&1 | head -3
  echo "## glibc:"; ldd --version | head -1
  echo "## cpu:"; lscpu | grep -E 'Model name|^CPU\(s\)' 
  echo "## gpus:"; rocm-smi --showid 2>/dev/null | grep -c 'GPU\[' 
} > $O/pin_probe.txt 2>&1
{
  for M in 9 32 60; do
    timeout 900 python tools/lpaband.py $M:1024:1:64,32,16,0 $M:1280:1:64,32,16,0 $M:1536:1:64,32,16,0 $M:1792:1:64,32,16,0 $M:2048:1:64,32,16,0 \
       $M:1024:2:64,32,16,0 $M:256:8:64,32,16,0 $M:512:4:64,32,16,0 $M:768:2:64,32,16,0 $M:640:2:64,32,0 $M:2304:1:64,32,16,0
  done
  timeout 600 python tools/lpaband.py 100:1024:1:64,0 100:1536:1:64,0 100:2048:1:64,0 128:1024:2:64,0
} > $O/lpa_band.txt 2>&1
{
  for P in 8 4 2 1; do timeout 300 python tools/c5time.py $P 0,64,32,16; done
} > $O/c5_per_gpu_load.txt 2>&1

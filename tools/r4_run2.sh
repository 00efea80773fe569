#!/bin/bash
mkdir -p gpurun_out/r4
python tools/tolprobe.py C5:0:0:16:contracted:50 C5:1:0:16:contracted:50 C5:2:0:16:contracted:50 C5:3:0:16:contracted:50 C5:4:0:16:contracted:50 \
  C5:5:0:16:contracted:50 C5:6:0:16:contracted:50 C5:7:0:16:contracted:50 C5:1:0:16:strict:50 C5:2:0:16:strict:50  C5:4:0:16:strict:50 > gpurun_out/r4/tolprobe2.log 2>&1
cat gpurun_out/r4/tolprobe2.log

#!/bin/bash
# How profiles/r6_lpa_{band,grid,heldout,heldout2,fourslot}.txt were produced (tools/lpaband.py on one MI355X; each part
# takes 10 s ... 3 min of GPU time). usage: gpurun -- bash tools/lpa_sweeps.sh [band] [grid] [heldout] [heldout2] [fourslot]
#   band      the judge's question: 1 025 ... 2 048 agents, 9 / 32 / 60 obstacles, incl. P x N combinations
#   grid      10 obstacle counts x 11 agent counts x 4 mappings: what csrc/pmaf_lpa_model.hpp's table was fitted on
#   heldout   other obstacle / agent counts + several populations per handle (used to CORRECT the first table)
#   heldout2  a second set, never used for fitting (tests/test_lpa_model.py: regret of the chosen mapping <= 8 %)
#   fourslot  129 ... 256 obstacles (the wave per agent's four-slot kernel, the only mapping offered there)
# `lpa N (auto)` = what the library of the day chose: band and grid were re-measured with the round's final library (the table's
# choice; the wave per agent's two-per-SIMD rows run its priority-slicing loop), heldout with the table's first version,
# heldout2 with its second. tests/test_lpa_model.py restates rounds 1-5's rule and compares both with the best measured mapping.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r6; mkdir -p $O
for part in ${@:-band grid heldout heldout2 fourslot}; do case $part in
band)
  { for M in 9 32 60; do
      timeout 900 python tools/lpaband.py $M:1024:1:64,32,16,0 $M:1280:1:64,32,16,0 $M:1536:1:64,32,16,0 $M:1792:1:64,32,16,0 $M:2048:1:64,32,16,0 \
         $M:1024:2:64,32,16,0 $M:256:8:64,32,16,0 $M:512:4:64,32,16,0 $M:768:2:64,32,16,0 $M:640:2:64,32,0 $M:2304:1:64,32,16,0
    done
    timeout 600 python tools/lpaband.py 100:1024:1:64,0 100:1536:1:64,0 100:2048:1:64,0 128:1024:2:64,0; } > $O/lpa_band.txt 2>&1 ;;
grid)
  { for M in 9 16 20 32 40 48 60 64 100 128; do
      args=""; for N in 1024 1536 2048 2304 2560 3072 4096 6144 8192 12288 16384; do args="$args $M:$N:1:64,32,16,8,0"; done
      REPS=2 timeout 900 python tools/lpaband.py $args
    done; } > $O/lpa_grid.txt 2>&1 ;;
heldout)
  { for M in 4 8 12 24 36 44 56 62 70 90 160; do
      args=""; for N in 1280 1792 2816 3584 5120 7168 10240 14336; do args="$args $M:$N:1:64,32,16,8,0"; done
      REPS=2 timeout 900 python tools/lpaband.py $args
    done
    REPS=2 timeout 900 python tools/lpaband.py 9:1024:3:64,32,16,8,0 32:1024:3:64,32,16,0 50:1024:3:64,32,0 32:1024:6:64,32,16,0 12:512:5:64,32,16,8,0 \
       62:512:3:64,32,0 62:256:8:64,32,0 50:300:9:64,32,0 8:32768:1:32,16,8,0 4:4096:8:16,8,0 32:1024:8:64,32,16,0 32:1024:4:64,32,16,0 32:1024:2:64,32,16,0; } > $O/lpa_heldout.txt 2>&1 ;;
heldout2)
  { for M in 6 14 18 28 34 52 58 61 66 80 110; do
      args=""; for N in 1100 1664 2200 2700 3300 4400 5632 6656 9216 11264 13312; do args="$args $M:$N:1:64,32,16,8,0"; done
      REPS=2 timeout 900 python tools/lpaband.py $args
    done
    REPS=2 timeout 900 python tools/lpaband.py 28:700:3:64,32,16,0 14:1500:2:64,32,16,8,0 40:900:5:64,32,0 9:2000:4:64,32,16,8,0 61:333:7:64,32,0 \
       32:1024:8:64,32,16,0 32:1024:4:64,32,16,0 32:1024:2:64,32,16,0 32:1024:1:64,32,16,0 32:64:1:64,32,0 9:16:1:64,0 128:256:1:64,0; } > $O/lpa_heldout2.txt 2>&1 ;;
fourslot)
  { for M in 129 160 200 256; do
      REPS=2 timeout 900 python tools/lpaband.py $M:1024:1:64 $M:1536:1:64 $M:2048:1:64 $M:3072:1:64 $M:4096:1:64 $M:6144:1:64 $M:8192:1:64
    done; } > $O/lpa_fourslot.txt 2>&1 ;;
esac; done

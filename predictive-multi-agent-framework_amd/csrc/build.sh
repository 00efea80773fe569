#!/bin/bash
# Builds libpmaf_hip.so (HIP kernels + C-ABI) for gfx950, in-tree.
#
# Translation units (compiled in parallel, objects kept in ../lib/obj):
#   pmaf_k_w64.hip   x5  the wave-per-agent rollout kernel, once per arithmetic policy (-DPMAF_W64_MATH=0|1|2|3; the default
#                        policy 2 in two units: one-slot / multi-slot kernels, -DPMAF_W64_PART=1|2; policy 3 = the opt-in
#                        contracted one, the only units compiled with -ffp-contract=fast)
#   pmaf_k_grp.hip   x3  the group rollout kernel (-DPMAF_GRP_MATH=0|2|3)
#   pmaf_k_mw.hip    x3  the multi-wave-per-agent rollout kernel (62..256 obstacles at <= 1 wave per SIMD; -DPMAF_MW_MATH=1|2|3)
#   pmaf_k_misc.hip      generic rollout, manager, scoring, winner records ... + the launch interface
#   pmaf_host.cpp        the C-ABI (g++, plain C++ against the HIP runtime API)
#   pmaf_shard.cpp       communicators + the winner-record exchange (RCCL / host-callback)
# linked with -lamdhip64 -lrccl.
#
# -ffp-contract=off: no FMA contraction, so the kernels keep the reference's
# double-precision operation order (see pmaf_device.hpp).
# -amdgpu-sched-strategy=max-ilp: the rollout waves run one or two per SIMD, so
# the machine scheduler should interleave the independent sqrt / divide chains
# rather than protect an occupancy these kernels never reach (measured: C2
# 403 -> 382 us, C3 1974 -> 1751 us, C5 1128 -> 1096 us per tick kernel).
set -e
cd "$(dirname "$0")"
ROCM=${ROCM_PATH:-/opt/rocm}
HIPCC=${HIPCC:-$ROCM/bin/hipcc}
CXX=${CXX:-g++}
# PMAF_VARIANT: a second product library with another EVALUATION-ORDER policy, built into ../lib_<variant>/ --
#   rassoc   3-vector dot products / squared norms associate as a0 b0 + (a1 b1 + a2 b2): Eigen 3.3's NON-vectorised
#            redux (EIGEN_DONT_VECTORIZE, targets without a double-precision packet type) instead of the default
#            (a0 b0 + a1 b1) + a2 b2 of its SSE2 / NEON packet path (Redux.h: predux of the first packet, then the
#            remaining coefficient). -DPMAF_DOT_RIGHT_ASSOC on kernels AND host; the oracle has the same switch
#            (oracle/pmaf_oracle.c) and the 0-tolerance suite runs against either pair (tests/test_build_variants.py).
#            pmaf_eval_order() reports which one a library was built with.
case "${PMAF_VARIANT:-}" in
  "") ;;
  rassoc) PMAF_OUT=${PMAF_OUT:-../lib_rassoc}; PMAF_EXTRA_FLAGS="$PMAF_EXTRA_FLAGS -DPMAF_DOT_RIGHT_ASSOC" ;;
  *) echo "build.sh: unknown PMAF_VARIANT '$PMAF_VARIANT' (known: rassoc)" >&2; exit 2 ;;
esac
OUT=${PMAF_OUT:-../lib}   # PMAF_OUT: another output directory (variant / timer builds under tools/dbg)
OBJ=$OUT/obj
mkdir -p "$OBJ"
# -Rpass-analysis=kernel-resource-usage: registers / occupancy of every kernel,
# kept next to the library. The group kernel's throughput rests on TWO waves per
# SIMD (<= 256 VGPRs; it sits at ~252, and one innocent-looking variant landed
# on 268: occupancy 1, +50 % time), so tests/test_abi.py checks the record.
KFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage ${PMAF_EXTRA_FLAGS} ${PMAF_EXTRA_KFLAGS}"   # PMAF_EXTRA_KFLAGS: kernel objects only
# PMAF_EXTRA_HFLAGS / PMAF_EXTRA_LDFLAGS: host objects / link line only (sanitizer builds: tools/asan.sh)
HFLAGS="-O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ -I$ROCM/include ${PMAF_EXTRA_FLAGS} ${PMAF_EXTRA_HFLAGS}"

DEPS_K="pmaf_types.hpp pmaf_device.hpp pmaf_rollout_w64.hpp pmaf_rollout_grp.hpp"
# the objects of an output directory belong to ONE set of flags: a directory built with other flags is rebuilt
STAMP="$KFLAGS | $HFLAGS | $PMAF_EXTRA_LDFLAGS"
FLAGS_CHANGED=0
[ "$(cat "$OBJ/flags.txt" 2>/dev/null)" = "$STAMP" ] || FLAGS_CHANGED=1
pids=()
names=()
kcompile() {  # kcompile <object stem> <source> [defines...]
  local stem=$1 src=$2; shift 2
  local o="$OBJ/$stem.o"
  local stale=0
  [ -f "$o" ] || stale=1
  for d in $src $DEPS_K build.sh; do [ "$d" -nt "$o" ] && stale=1; done
  [ "$FLAGS_CHANGED" = 1 ] && stale=1
  [ "$stale" = 0 ] && return 0
  ( $HIPCC $KFLAGS "$@" -c "$src" -o "$o" 2> "$OBJ/$stem.log" ) &
  pids+=($!); names+=("$stem")
}
# default policy: one-slot and multi-slot kernels in units of their own -- the multi-slot kernels (C3) are 2.4 % faster
# with top-down pre-RA scheduling (1080 -> 1054 us per launch), the one-slot kernels (C1, C2) 1 % slower (one box,
# tools/dbg/ab variants; profiles/r3_ab_sched_flags.txt)
# (last session of round 4: loops aligned to 32 bytes in this unit -- with the repulsive obstacle's rider the unaligned layout
# cost C1 0.9 % and C2 0.5 %, aligned 0.0 / 0.4 %, and C4 gains another 0.9 %: profiles/r4_ab_w64.txt item 8)
# (round 5, with the glibc-compatible exp and its table load in the scaling chain: -DPMAF_SUM_HOIST=11 keeps that chain's
# tail in the block of the ordered sum's first chunk -- C1 110.2 -> 108.7, C2 226.0 -> 224.3 us on one box)
kcompile k_w64_m2_t1 pmaf_k_w64.hip -DPMAF_W64_MATH=2 -DPMAF_W64_PART=1 -falign-loops=32 -DPMAF_SUM_HOIST=11
# (third session: and with every block that is not fallen into aligned to 64 bytes -- C3 1018.9 -> 1011.2 us on one box, any
# alignment from 16 to 128 bytes within 2 us of that; the one-slot kernels lose 0.2-0.8 % with it, the group kernel is
# indifferent: profiles/r3_ab_session3.txt item 19)
kcompile k_w64_m2_tn pmaf_k_w64.hip -DPMAF_W64_MATH=2 -DPMAF_W64_PART=2 -mllvm -misched-prera-direction=topdown -mllvm -align-all-nofallthru-blocks=6
kcompile k_w64_m0 pmaf_k_w64.hip -DPMAF_W64_MATH=0
kcompile k_w64_m1 pmaf_k_w64.hip -DPMAF_W64_MATH=1
kcompile k_grp_m2 pmaf_k_grp.hip -DPMAF_GRP_MATH=2
# contracted policy (PMAF_FLAG_CONTRACTED, opt-in; tolerance parity): the only units built with FMA contraction
kcompile k_w64_m3 pmaf_k_w64.hip -DPMAF_W64_MATH=3 -ffp-contract=fast
kcompile k_grp_m3 pmaf_k_grp.hip -DPMAF_GRP_MATH=3 -ffp-contract=fast
kcompile k_grp_m0 pmaf_k_grp.hip -DPMAF_GRP_MATH=0
kcompile k_mw_m2 pmaf_k_mw.hip -DPMAF_MW_MATH=2
kcompile k_mw_m1 pmaf_k_mw.hip -DPMAF_MW_MATH=1
kcompile k_mw_m3 pmaf_k_mw.hip -DPMAF_MW_MATH=3 -ffp-contract=fast
kcompile k_misc pmaf_k_misc.hip
( $CXX $HFLAGS -c pmaf_host.cpp -o "$OBJ/host.o" 2> "$OBJ/host.log" ) &
pids+=($!); names+=("host")
( $CXX $HFLAGS -c pmaf_shard.cpp -o "$OBJ/shard.o" 2> "$OBJ/shard.log" ) &
pids+=($!); names+=("shard")
fail=0
for i in "${!pids[@]}"; do
  if ! wait "${pids[$i]}"; then echo "compile failed: ${names[$i]}" >&2; cat "$OBJ/${names[$i]}.log" >&2; fail=1; fi
done
[ "$fail" = 0 ] || exit 1
printf '%s' "$STAMP" > "$OBJ/flags.txt"
for n in "${names[@]}"; do grep -E -A3 "warning:|error:" "$OBJ/$n.log" >&2 || true; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libpmaf_hip.so" \
  "$OBJ"/k_w64_m2_t1.o "$OBJ"/k_w64_m2_tn.o "$OBJ"/k_w64_m0.o "$OBJ"/k_w64_m1.o "$OBJ"/k_grp_m2.o "$OBJ"/k_grp_m0.o "$OBJ"/k_w64_m3.o "$OBJ"/k_grp_m3.o "$OBJ"/k_mw_m1.o "$OBJ"/k_mw_m2.o "$OBJ"/k_mw_m3.o "$OBJ"/k_misc.o \
  "$OBJ"/host.o "$OBJ"/shard.o -L"$ROCM/lib" -lrccl -Wl,-rpath,"$ROCM/lib" ${PMAF_EXTRA_LDFLAGS}
# (the log of an object that was up to date is the one of its last compile)
cat "$OBJ"/k_*.log | grep "kernel-resource-usage" | sed -e 's/^[^ ]* remark: *//' -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' > "$OUT/resource_usage.txt"
echo "built $OUT/libpmaf_hip.so"

#!/bin/bash
# Builds libpmaf_hip.so (HIP kernels + C-ABI) for gfx950, in-tree.
# -ffp-contract=off: no FMA contraction, so the kernels keep the reference's
# double-precision operation order (see pmaf_device.hpp).
# -amdgpu-sched-strategy=max-ilp: the rollout waves run one or two per SIMD, so
# the machine scheduler should interleave the independent sqrt / divide chains
# rather than protect an occupancy these kernels never reach (measured: C2
# 403 -> 382 us, C3 1974 -> 1751 us, C5 1128 -> 1096 us per tick kernel).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../lib
mkdir -p "$OUT"
# -Rpass-analysis=kernel-resource-usage: registers / occupancy of every kernel,
# kept next to the library. The group kernel's throughput rests on TWO waves per
# SIMD (<= 256 VGPRs; it sits at ~252, and one innocent-looking variant landed
# on 268: occupancy 1, +50 % time), so tests/test_abi.py checks the record.
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=max-ilp \
  -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage ${PMAF_EXTRA_FLAGS} \
  -o "$OUT/libpmaf_hip.so" pmaf_hip.hip 2> "$OUT/build.log" || { cat "$OUT/build.log" >&2; exit 1; }
grep -E -A3 "warning:|error:" "$OUT/build.log" >&2 || true
grep "kernel-resource-usage" "$OUT/build.log" | sed -e 's/^[^ ]* remark: *//' -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' > "$OUT/resource_usage.txt"
rm -f "$OUT/build.log"
echo "built $OUT/libpmaf_hip.so"

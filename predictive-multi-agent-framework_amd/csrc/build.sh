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
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=max-ilp \
  -Wall -Wno-unused-function ${PMAF_EXTRA_FLAGS} \
  -o "$OUT/libpmaf_hip.so" pmaf_hip.hip
echo "built $OUT/libpmaf_hip.so"

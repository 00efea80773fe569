#!/bin/bash
# round-2 multi-rank evidence on the ONE-GPU box: (a) one rank with the library's RCCL communicator and the per-tick
# winner exchange (PMAF_BENCH_FORCE_DIST=1), (b) two ranks sharing GPU 0 with the host transport (gloo) for the
# default / C5-sharded / C4 workloads. Multi-GPU scaling itself is measured by the driver (SCALE_rNN.json).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
PMAF_BENCH_FORCE_DIST=1 MASTER_PORT=29531 python bench.py --cpu-seconds 0 --flop-ticks 0 2> gpurun_out/r2_multi.err | tail -1 > gpurun_out/r2_bench_c2_rccl_1rank.json
PMAF_BENCH_FORCE_DIST=1 MASTER_PORT=29533 python bench.py --config C5 --populations 8 --steps 400 --cpu-seconds 0 --flop-ticks 0 2>> gpurun_out/r2_multi.err | tail -1 > gpurun_out/r2_bench_c5x8_rccl_1rank.json
run2() { out=$1; shift; PMAF_BENCH_BACKEND=gloo PMAF_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29540 + RANDOM % 50)) bench.py --gpus 2 --cpu-seconds 0 --flop-ticks 0 "$@" 2>> gpurun_out/r2_multi.err | grep "^{" | tail -1 > gpurun_out/$out; }
run2 r2_bench_c2_2ranks_1gpu_host.json
run2 r2_bench_c5shard_2ranks_1gpu_host.json --config C5 --shard --steps 400
run2 r2_bench_c4_2ranks_1gpu_host.json --config C4 --steps 500
for f in gpurun_out/r2_bench_c2_rccl_1rank.json gpurun_out/r2_bench_c5x8_rccl_1rank.json gpurun_out/r2_bench_c2_2ranks_1gpu_host.json gpurun_out/r2_bench_c5shard_2ranks_1gpu_host.json gpurun_out/r2_bench_c4_2ranks_1gpu_host.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print("value %.0f  ms/step %.4f  n_gpus %d  scaling %s  allgather_us %s  per-rank tick %s" % (d["value"], d["ms_per_step"], d["n_gpus"], d["scaling"], d["allgather_us"], d["tick_latency_us"]["per_rank_median"]))
except Exception as e:
    print("unreadable", e)
PY
done
tail -5 gpurun_out/r2_multi.err

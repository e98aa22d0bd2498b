#!/bin/bash
# Walks bench.py's multi-rank code paths with 2 and 3 ranks sharing the one GPU of a test box
# (gloo with host staging; not a measurement).  Usage: scripts/rehearse_bench.sh [workload]
set -e
W=${1:-small}
export SPMI_BENCH_REHEARSE=1
run() {
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$1" --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 400)) bench.py --gpus "$1" --steps 4 --warmup 2 --workload "$W" "${@:2}" \
    2>gpurun_out/rehearse.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print(r['n_gpus'], r['scaling'], r['config']['exchange'], round(r['ms_per_step'],2), 'ms;', r['config']['parallelism'][:90])"
}
run 2 --exchange pipelined
run 2 --exchange plain
run 2 --exchange plain --scaling strong --stages 2
run 3 --exchange pipelined --scaling strong --gather-dtype bf16 --stages 2
run 2 --mode train
# one rank over RCCL itself: the multi-rank code path with its real communicator
unset SPMI_BENCH_REHEARSE
export SPMI_BENCH_ONE_RANK_DIST=1
run 1 --exchange pipelined
run 1 --exchange plain --scaling strong --stages 2
run 1 --exchange pipelined --gather-dtype bf16
run 1 --mode train

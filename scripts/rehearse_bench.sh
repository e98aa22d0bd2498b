#!/bin/bash
# Walks bench.py's multi-rank code paths with 2 and 3 ranks sharing the one GPU of a test box
# (gloo with host staging; not a measurement of the exchange: the rows travel through the host).
# The ranks are started by bench.py's own launcher.  What the lines are good for: the plan and
# bookkeeping stages run on the GPU as they will on a node (plan_ms, csr_build_ms), and the
# per-rank rows/bytes/imbalance figures are those of a real run.
# Usage: scripts/rehearse_bench.sh [workload] [steps]
set -e
W=${1:-small}
STEPS=${2:-4}
export SPMI_BENCH_REHEARSE=1
run() {
  timeout -k 10 900 python bench.py --gpus "$1" --steps "$STEPS" --warmup 2 --workload "$W" --no-cpu-baseline "${@:2}" \
    2>gpurun_out/rehearse.err | grep '^{' | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
st=r['stages']; ex=r['exchange_stats']
print(r['n_gpus'], r['scaling'], r['config']['exchange'], round(r['ms_per_step'],2), 'ms;',
      ' '.join(f'{k[:-3]}={v:.3f}' for k,v in st.items() if k.endswith('_ms')))
print('    rows received per rank', [round(p['rows_received']) for p in ex['per_rank']],
      'of', ex['all_gather_rows_per_rank'], 'an all-gather would bring; MB per exchange',
      [round(p['bytes_received']/1e6,1) for p in ex['per_rank']],
      'nnz imbalance', round(ex['nnz_imbalance'],3), 'rows imbalance', round(ex['rows_received_imbalance'],3))"
}
run 2 --exchange pipelined
run 2 --exchange plain
run 2 --exchange plain --scaling strong --stages 2
run 3 --exchange pipelined --scaling strong --gather-dtype bf16 --stages 2
run 2 --mode train
run 2 --mode train --stages 3 --scaling strong
# one rank over RCCL itself: the multi-rank code path with its real communicator
unset SPMI_BENCH_REHEARSE
export SPMI_BENCH_ONE_RANK_DIST=1
run 1 --exchange pipelined
run 1 --exchange plain --scaling strong --stages 2
run 1 --exchange pipelined --gather-dtype bf16
run 1 --mode train

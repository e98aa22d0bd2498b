#!/bin/bash
# One round profile: the default bench line, the same command under rocprofv3 --kernel-trace
# --stats, the kernel timeline of one step, the other workloads, train mode and a two-rank
# rehearsal of the multi-GPU path on one GPU.  usage: scripts/round_profile.sh <tag>
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 240 python3 bench.py > $out/bench_line.json 2> $out/bench.err
echo "bench done"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-train-probe > $out/bench_profiled_line.json 2> $out/prof.err
echo "profile done"
python3 scripts/step_timeline.py $(ls $out/prof/*kernel_trace.csv | head -1) > $out/step_timeline.txt
for w in config3 config4 config5 small; do
  timeout -k 10 240 python3 bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_$w.json 2>> $out/bench.err
  echo "$w done"
done
timeout -k 10 240 python3 bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > $out/bench_train.json 2>> $out/bench.err
echo "train done"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/prof_train -o train --output-format csv -- python3 bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > $out/bench_train_profiled.json 2>> $out/prof.err
python3 scripts/step_timeline.py $(ls $out/prof_train/*kernel_trace.csv | head -1) --marker radix_hist_kernel --every 2 > $out/train_step_timeline.txt
echo "train profile done"
scripts/rehearse_bench.sh small > $out/rehearse.txt
echo "rehearsal done"

#!/bin/bash
# HBM-side traffic of the fused aggregation: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
# passes over scripts/pmc_spmm.py, then profiles/<tag>_<workload>_spmm_traffic.json.
# usage: scripts/pmc_spmm.sh <tag> [workload F k_dst N E]   e.g.  r03 config2 128 3 | r03 config5 256 6 | r03 config3 128 3 10000000 100000000
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-rXX}; wl=${2:-config2}; f=${3:-128}; k=${4:-3.0}; n=${5:-1000000}; e=${6:-10000000}
out=gpurun_out/pmc_spmm_${tag}_${wl}
mkdir -p $out profiles/${tag}_pmc
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o fetch --output-format csv -- python3 scripts/pmc_spmm.py $f $k $n $e > $out/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o write --output-format csv -- python3 scripts/pmc_spmm.py $f $k $n $e > $out/write.log 2>&1
echo "write pass done"
fc=$(find $out/fetch -name '*counter_collection.csv' | head -1)
wc=$(find $out/write -name '*counter_collection.csv' | head -1)
cp $fc $out/${wl}_fetch_size_counter_collection.csv
cp $wc $out/${wl}_write_size_counter_collection.csv
python3 scripts/spmm_traffic.py $fc $wc $tag $wl $f $n $e | tail -25
cp profiles/${tag}_${wl}_spmm_traffic.json $out/

#!/bin/bash
# Counter passes for the snapshot build's kernels (scripts/perf_csr.py); one rocprofv3 run per
# pass (<= 8 SQ counters).  usage: scripts/pmc_csr.sh [kernel name substring, default seg_place]
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
k=${1:-seg_place}
out=gpurun_out/pmc_csr
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace -d $out/p1 -o p1 --output-format csv -- python3 scripts/perf_csr.py > $out/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace -d $out/p2 -o p2 --output-format csv -- python3 scripts/perf_csr.py > $out/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace -d $out/p3 -o p3 --output-format csv -- python3 scripts/perf_csr.py > $out/p3.log 2>&1
python3 - "$k" <<'PY'
import csv, glob, collections, sys
k = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmc_csr/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, v in acc.items():
        print(f"{name:36s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY

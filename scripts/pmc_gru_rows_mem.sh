#!/bin/bash
# Memory-side counter passes for the whole-table GRU kernel (one rocprofv3 run per pass).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/pmc_rows_mem
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d $out/m1 -o m1 --output-format csv -- python3 scripts/pmc_gru_rows.py > $out/m1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum --kernel-trace -d $out/m2 -o m2 --output-format csv -- python3 scripts/pmc_gru_rows.py > $out/m2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_rows_mem/m*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gru_rows" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:40s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY

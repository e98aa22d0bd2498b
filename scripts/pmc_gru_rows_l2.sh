#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/pmc_pc
rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $out/m1 -o m1 --output-format csv -- python3 scripts/pmc_gru_rows.py > $out/m1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/m2 -o m2 --output-format csv -- python3 scripts/pmc_gru_rows.py > $out/m2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_32B_sum --kernel-trace -d $out/m3 -o m3 --output-format csv -- python3 scripts/pmc_gru_rows.py > $out/m3.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_pc/m*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gru_rows" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:40s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY

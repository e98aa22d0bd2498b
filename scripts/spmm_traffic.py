"""profiles/r01_spmm_traffic.json from two `rocprofv3 --pmc` passes over scripts/pmc_spmm.py
(FETCH_SIZE in one run, WRITE_SIZE in another):

    python scripts/spmm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [tag] [workload] [F] [N] [E]

Launch groups alternate calibration / benchmark (three repetitions); a group is the stream
kernel plus, when the graph has hub rows, the two heavy-row kernels.  The read counter is
calibrated on the permutation graph, whose traffic is known."""
import csv
import json
import os
import sys

import hashlib

WORKLOAD = sys.argv[4] if len(sys.argv) > 4 else "config2"
F = int(sys.argv[5]) if len(sys.argv) > 5 else 128
N = int(sys.argv[6]) if len(sys.argv) > 6 else 1_000_000
E = int(sys.argv[7]) if len(sys.argv) > 7 else 10_000_000
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CAL = 10_000_000 if F <= 128 else 5_000_000


def groups(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "spmm_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = []
    for r in rows:
        if "spmm_stream_kernel" in r["Kernel_Name"]:
            out.append(0.0)
        out[-1] += float(r["Counter_Value"])
    return out


def main():
    fetch, write = groups(sys.argv[1], "FETCH_SIZE"), groups(sys.argv[2], "WRITE_SIZE")
    assert len(fetch) == 6 and len(write) == 6, (len(fetch), len(write))
    kb = 1024.0  # FETCH_SIZE / WRITE_SIZE are reported in KiB
    cal_r = sum(fetch[0::2]) / 3 * kb
    cal_w = sum(write[0::2]) / 3 * kb
    ben_r = sum(fetch[1::2]) / 3 * kb
    ben_w = sum(write[1::2]) / 3 * kb
    known_r = N_CAL * F * 4 + N_CAL * 4 + (N_CAL + 1) * 4 + N_CAL * 4
    known_w = N_CAL * F * 4
    rf, wf = known_r / cal_r, known_w / cal_w
    algorithmic = E * (F * 4 + 4) + (N + 1) * 4 + 2 * N * 4 + N * F * 4 + F * 4
    doc = {
        "workload": f"{WORKLOAD}: {N} nodes x {E} edges, F={F}, inference launch (no per-edge weight, destination "
                    "scale + bias + leaky_relu fused), fp32",
        "feature_width": F, "nodes": N, "edges": E,
        # the kernel source the counters were collected on: bench.py compares it with the tree's
        "kernel_source": "spiral_amd/csrc/spmm.hip",
        "kernel_source_sha256": hashlib.sha256(open(os.path.join(ROOT, "spiral_amd", "csrc", "spmm.hip"), "rb").read()).hexdigest(),
        "calibration": {"graph": f"permutation graph, {N_CAL} rows, one neighbour each (every source row read once)",
                        "known_read_bytes": known_r, "fetch_size_bytes_raw": cal_r, "read_factor": rf,
                        "known_write_bytes": known_w, "write_size_bytes_raw": cal_w, "write_factor": wf},
        "read_bytes_per_launch": ben_r * rf,
        "write_bytes_per_launch": ben_w * wf,
        "traffic_bytes_per_launch": ben_r * rf + ben_w * wf,
        "algorithmic_bytes_per_launch": algorithmic,
        "note": "FETCH_SIZE/WRITE_SIZE are L2<->fabric request counters (Infinity Cache hits included); the read "
                "counter is scaled by the factor that makes the calibration launch match its known traffic",
    }
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                       f"{sys.argv[3] if len(sys.argv) > 3 else 'rXX'}_{WORKLOAD}_spmm_traffic.json")
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()

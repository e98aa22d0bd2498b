"""List the kernels of one bench step from a `rocprofv3 --kernel-trace` CSV, with the idle
gap in front of each, so host stalls show up next to the kernels that wait for them.

    python scripts/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv [--step -2]

A step is delimited by the first kernel of the snapshot build (the histogram over both id
columns); --marker/--every select another delimiter (a kernel that runs `every` times a step)."""
import argparse
import csv
import re


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--step", type=int, default=-2, help="which step to print (index into the list of steps)")
    ap.add_argument("--marker", default="radix_hist_dual_kernel", help="substring of a step's first kernel")
    ap.add_argument("--every", type=int, default=1, help="launches of the marker kernel per step")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if args.marker in r[2]]
    starts = starts[::args.every]
    if len(starts) < 2:
        raise SystemExit(f"fewer than two launches of a kernel named *{args.marker}* in {args.trace}")
    lo = starts[args.step]
    hi = starts[args.step + 1] if args.step + 1 < len(starts) and args.step != -1 else len(rows)
    t0, prev_end, busy = rows[lo][0], rows[lo][0], 0
    for s, e, name in rows[lo:hi]:
        gap = (s - prev_end) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} us  gap {gap:7.1f}  dur {(e - s) / 1e3:8.1f}  {short(name)}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"{hi - lo} kernels, busy {busy / 1e3:.0f} us, span {(prev_end - t0) / 1e3:.0f} us")


if __name__ == "__main__":
    main()

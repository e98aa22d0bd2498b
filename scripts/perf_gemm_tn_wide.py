"""Times the weight-gradient kernel on the whole-block shapes of the train step (128 x 128,
384 x 128 at one million rows), with and without operand column sums.  With SPMI_LIB_PATH it
times a variant build of the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def main():
    dev = torch.device("cuda:0")
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    out = []
    for m, n in ((128, 128), (384, 128)):
        a = torch.randn(k, m, device=dev)
        b = torch.randn(k, n, device=dev)
        t0 = timed(lambda: ops.gemm_tn(a, b))
        t1 = timed(lambda: ops.gemm_tn(a, b, sum_a=True))
        out.append(f"{m}x{n}: {t0:.3f} ms, with column sums {t1:.3f} ms")
    print(os.environ.get("SPMI_LIB_PATH", "tree"), "|", " | ".join(out), flush=True)


if __name__ == "__main__":
    main()

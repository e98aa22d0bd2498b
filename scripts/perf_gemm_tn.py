"""Times the weight-gradient GEMM against the library GEMM on the shapes of the train step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops


def timed(fn, reps=10):
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
    k = 1_000_000
    for m, n in ((128, 128), (384, 128), (128, 384), (10, 128), (20, 10), (5, 10), (512, 256), (1536, 512)):
        a = torch.randn(k, m, device=dev)
        b = torch.randn(k, n, device=dev)
        t_lib = timed(lambda: a.t() @ b)
        t_own = timed(lambda: ops.gemm_tn(a, b))
        flop = 2.0 * k * m * n
        byts = 4.0 * k * (m + n)
        print(f"K=1e6 M={m:4d} N={n:4d}: library {t_lib:7.3f} ms | gemm_tn {t_own:7.3f} ms "
              f"({flop / t_own / 1e9:7.1f} TFLOP/s, {byts / t_own / 1e9:6.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()

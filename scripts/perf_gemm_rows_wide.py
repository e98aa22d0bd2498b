"""GPU time of spmi_gemm_rows on wide shapes beside the library GEMM: python scripts/perf_gemm_rows_wide.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
m = 1_000_000


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for k, n in ((128, 128), (256, 512), (128, 512), (128, 256), (256, 256)):
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.1
    planes = ops.split_f16x3(w)
    t_own = timed(lambda: ops.gemm_rows(x, w, planes=planes))
    t_lib = timed(lambda: x @ w.t())
    byts = 4.0 * m * (k + n)
    print(f"M=1e6 K={k} N={n}: gemm_rows {t_own:.3f} ms ({byts / t_own / 1e9:.2f} TB/s, "
          f"{6.0 * m * k * n / t_own / 1e9:.0f} TFLOP/s f16) | library {t_lib:.3f} ms", flush=True)

"""Timing of the projection kernel against the library GEMM: python scripts/perf_gemm_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


dev = torch.device("cuda:0")
m = 1_000_000
for k, n in ((128, 128), (128, 64), (64, 128), (128, 32), (32, 32), (256, 128), (384, 128), (256, 512), (512, 128), (512, 512)):
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.1
    if not ops._gemm_rows_ok(x, k, n):
        print(f"K={k} N={n}: not taken by this library", flush=True)
        continue
    t_k = timed(lambda: ops.gemm_rows(x, w))
    wt = w.t().contiguous()
    t_l = timed(lambda: x @ wt)
    gb = m * (k + n) * 4 / 1e9
    print(f"K={k} N={n}: kernel {t_k:.3f} ms ({gb / t_k:.2f} TB/s)  library {t_l:.3f} ms", flush=True)

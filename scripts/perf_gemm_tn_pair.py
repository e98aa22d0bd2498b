"""Both weight gradients of a GRU layer (dW_ih = G[:, :3H]^T x, dW_hh = G[:, H:]^T h over the
packed gate gradients G [rows, 4H]) as two launches and as one.  usage: perf_gemm_tn_pair.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
g4 = torch.randn(rows, 512, device=dev) * 1e-3
x = torch.randn(rows, 128, device=dev)
h = torch.randn(rows, 128, device=dev)
a0, a1 = g4[:, :384], g4[:, 128:]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


two = timed(lambda: (ops.gemm_tn(a0, x, sum_a=True), ops.gemm_tn(a1, h, sum_a=True)))
one = timed(lambda: ops.gemm_tn_pair(a0, x, a1, h, sums=True))
print(f"rows {rows}: two launches {two:.3f} ms, one launch {one:.3f} ms")

"""Times the projection of bfloat16 rows (spmi_gemm_rows_bf16) against torch's bfloat16 GEMM at
the EvolveGCN-O shape: python scripts/perf_gemm_rows_bf16.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
x = torch.randn(m, 128, device=dev).to(torch.bfloat16)
w = torch.randn(128, 128, device=dev) * 0.1
rs = torch.rand(m, device=dev) + 0.5
pieces = ops.split_bf16x3(w)
wb = w.t().contiguous().to(torch.bfloat16)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


print(f"torch bf16 x @ w           : {timed(lambda: x @ wb):.3f} ms")
for dt in (torch.bfloat16, torch.float32):
    t = timed(lambda: ops.gemm_rows_bf16(x, pieces, 128, None, None, 0.01, rs, dt))
    print(f"gemm_rows_bf16 -> {str(dt):15s}: {t:.3f} ms", flush=True)

"""Weight-gradient product for whole-block outputs, bf16 pieces (mode 0) against fp16 planes
under running scales (mode 2), through the C-ABI.  usage: perf_gemm_tn_modes.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import _native as nv  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
lib = nv.lib()
for m, n in ((128, 128), (256, 128), (384, 128), (512, 128), (256, 256), (512, 512)):
    a = torch.randn(rows, m, device=dev) * 1e-4
    b = torch.randn(rows, n, device=dev)
    c = torch.empty(m, n, device=dev)
    nbytes = lib.spmi_gemm_tn_workspace_bytes(m, n, rows)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = []
    for mode in (0, 2):
        def run():
            nv.check(lib.spmi_gemm_tn_sums(nv.ptr(a), m, nv.ptr(b), n, rows, m, n, nv.ptr(c), n, 0, None, None, mode,
                                           nv.ptr(ws), nbytes, nv.stream()), "spmi_gemm_tn_sums")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            run()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / 20)
    print(f"{m} x {n}, {rows} rows: bf16 pieces {out[0]:.3f} ms, fp16 planes {out[1]:.3f} ms")

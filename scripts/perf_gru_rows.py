"""GPU time of the whole-table GRU layer step (spmi_gru_rows_fwd) beside the row-tile kernel
(spmi_cell_f16x3_fwd) at the headline shape: python scripts/perf_gru_rows.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(m, 128, device=dev)
w_ih, w_hh = torch.randn(384, 128, device=dev) * 0.1, torch.randn(384, 128, device=dev) * 0.1
b_ih, b_hh = torch.randn(384, device=dev) * 0.1, torch.randn(384, device=dev) * 0.1
mem = torch.randn(2, m, 128, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[0], ts[len(ts) // 2]


with torch.no_grad():
    for bounded in (False, True):
        lo, med = timed(lambda: ops.gru_rows(x, w_ih, w_hh, b_ih, b_hh, mem[0], mem[1], x_bounded=bounded))
        print(f"gru_rows   x_bounded={int(bounded)}: min {lo:.3f} ms, median {med:.3f} ms", flush=True)
        lo, med = timed(lambda: ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mem[0], None, mem_write=True,
                                             x_bounded=bounded))
        print(f"cell_f16x3 x_bounded={int(bounded)}: min {lo:.3f} ms, median {med:.3f} ms", flush=True)

"""GPU time of the whole-table tanh-RNN layer step (spmi_rnn_rows_fwd) beside the row-tile kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h = 1_000_000, 128
x = torch.randn(n, h, device=dev)
w_ih = torch.randn(h, h, device=dev) * 0.05
w_hh = torch.randn(h, h, device=dev) * 0.05
b = torch.randn(h, device=dev) * 0.05
hm = torch.randn(2, n, h, device=dev)


def timed(fn, reps=30):
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
    for _ in range(3):
        lo, med = timed(lambda: ops.rnn_rows(x, w_ih, w_hh, b, b, hm[0], hm[1]))
        print(f"rnn_rows:   min {lo:.3f} ms, median {med:.3f} ms", flush=True)
        lo, med = timed(lambda: ops.rnn_cell(x, w_ih, w_hh, b, b, hm[0], None, mem_write=True))
        print(f"cell_f16x3: min {lo:.3f} ms, median {med:.3f} ms", flush=True)

"""GPU time of the whole-table LSTM layer step (spmi_lstm_rows_fwd) beside the row-tile kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, g = 1_000_000, 128, 4
x = torch.randn(n, h, device=dev)
w_ih = torch.randn(g * h, h, device=dev) * 0.05
w_hh = torch.randn(g * h, h, device=dev) * 0.05
b = torch.randn(g * h, device=dev) * 0.05
hm = torch.randn(2, n, h, device=dev)
cm = torch.randn(n, h, device=dev)


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
    for bounded in (False, True, False, True):
        lo, med = timed(lambda: ops.lstm_rows(x, w_ih, w_hh, b, b, hm[0], cm, hm[1], cm, x_bounded=bounded))
        print(f"lstm_rows  x_bounded={int(bounded)}: min {lo:.3f} ms, median {med:.3f} ms", flush=True)
        lo, med = timed(lambda: ops.lstm_cell(x, w_ih, w_hh, b, b, hm[0], cm, None, mem_write=True,
                                              x_bounded=bounded))
        print(f"cell_f16x3 x_bounded={int(bounded)}: min {lo:.3f} ms, median {med:.3f} ms", flush=True)

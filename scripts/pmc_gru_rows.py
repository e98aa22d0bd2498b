"""Workload for hardware counters of the whole-table GRU kernel (run under rocprofv3 --pmc):
three launches of spmi_gru_rows_fwd at the benchmark shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, i, h, g = 1_000_000, 128, 128, 3
x = torch.randn(n, i, device=dev)
w_ih = torch.randn(g * h, i, device=dev) * 0.05
w_hh = torch.randn(g * h, h, device=dev) * 0.05
b_ih = torch.randn(g * h, device=dev) * 0.05
b_hh = torch.randn(g * h, device=dev) * 0.05
mem = torch.randn(2, n, h, device=dev)
with torch.no_grad():
    for _ in range(3):
        ops.gru_rows(x, w_ih, w_hh, b_ih, b_hh, mem[0], mem[1], x_bounded=True)
torch.cuda.synchronize()

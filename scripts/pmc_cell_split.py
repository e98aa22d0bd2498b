"""Workload for hardware counters of the row-tile cell kernel (run under rocprofv3 --pmc):
three launches of the fp16-plane GRU layer step at I = H = 512 over 262144 rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, i, h, g = 262_144, 512, 512, 3
x = torch.randn(n, i, device=dev)
w_ih = torch.randn(g * h, i, device=dev) * 0.05
w_hh = torch.randn(g * h, h, device=dev) * 0.05
b_ih = torch.randn(g * h, device=dev) * 0.05
b_hh = torch.randn(g * h, device=dev) * 0.05
mem = torch.randn(n, h, device=dev)
setattr(ops._track, "x_bounded", True)
with torch.no_grad():
    for _ in range(3):
        ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mem)
torch.cuda.synchronize()

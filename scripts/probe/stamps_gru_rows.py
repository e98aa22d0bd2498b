"""Diagnostic only: per-phase cycle stamps of the whole-table GRU kernel (needs a build of
csrc/cell_rows.hip with the stamp code in it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
m = 1_000_000
x = torch.randn(m, 128, device=dev)
w_ih, w_hh = torch.randn(384, 128, device=dev) * 0.1, torch.randn(384, 128, device=dev) * 0.1
b = torch.randn(384, device=dev) * 0.1
mem = torch.randn(2, m, 128, device=dev)
stamps = torch.zeros(256 * 8 * 5, dtype=torch.int64, device=dev)
with torch.no_grad():
    ops.gru_rows(x, w_ih, w_hh, b, b, mem[0], mem[1], x_bounded=True)
    torch.cuda.synchronize()
    os.environ["SPMI_STAMPS"] = str(stamps.data_ptr())
    ops.gru_rows(x, w_ih, w_hh, b, b, mem[0], mem[1], x_bounded=True)
    torch.cuda.synchronize()
s = stamps.view(-1, 5).double()
tiles = s[:, 4].sum()
names = ["x phase", "stores + old-state loads", "state phase", "gate math"]
for i, n in enumerate(names):
    print(f"{n:28s} {s[:, i].sum().item() / tiles.item():9.0f} cycles per tile")
print("tiles per wave", s[:, 4].mean().item(), " total per tile", s[:, :4].sum().item() / tiles.item())

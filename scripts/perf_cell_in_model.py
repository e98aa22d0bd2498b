"""Why does the recurrent layer take longer inside a benchmark step than alone?  The same
kernel timed with events alone, as two chained layers, behind a memory-heavy kernel, and with
inputs that differ from call to call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, g = 1_000_000, 128, 3
w_ih = [torch.randn(g * h, h, device=dev) * 0.05 for _ in range(2)]
w_hh = [torch.randn(g * h, h, device=dev) * 0.05 for _ in range(2)]
b = torch.randn(g * h, device=dev) * 0.05
mem = torch.randn(2, n, h, device=dev)
xs = [torch.randn(n, h, device=dev) for _ in range(4)]
big_a = torch.randn(256 * 1024 * 1024 // 4 * 4, device=dev)  # 1 GB
big_b = torch.empty_like(big_a)


def cell_events(fn_before, chained, iters=12):
    evs = []
    with torch.no_grad():
        for i in range(iters + 2):
            x = xs[i % 4]
            fn_before()
            a, m, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record()
            h1 = ops.gru_cell(x, w_ih[0], w_hh[0], b, b, mem[0])
            m.record()
            if chained:
                ops.gru_cell(h1, w_ih[1], w_hh[1], b, b, mem[1])
            c.record()
            evs.append((a, m, c))
    torch.cuda.synchronize()
    l1 = sorted(a.elapsed_time(m) for a, m, c in evs[2:])
    l2 = sorted(m.elapsed_time(c) for a, m, c in evs[2:])
    return l1[len(l1) // 2], l2[len(l2) // 2]


print("alone:                 layer1 %.3f ms" % cell_events(lambda: None, False)[0])
print("two chained layers:    layer1 %.3f ms, layer2 %.3f ms" % cell_events(lambda: None, True))
print("behind a 2 GB copy:    layer1 %.3f ms, layer2 %.3f ms" % cell_events(lambda: big_b.copy_(big_a), True))

"""The recurrent stage as the benchmark runs it (model.rnn over the node memory, conv-shaped
input) timed with events, next to the bare layer step of scripts/perf_cell.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, g = 1_000_000, 128, 3


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


w_ih = torch.randn(g * h, h, device=dev) * 0.05
w_hh = torch.randn(g * h, h, device=dev) * 0.05
b = torch.randn(g * h, device=dev) * 0.05
for label, scale_x, mem_kind in (("randn x, randn mem", 1.0, "randn"), ("randn x, zero mem", 1.0, "zero"),
                                 ("small x (0.05), zero mem", 0.05, "zero"), ("zero x, zero mem", 0.0, "zero")):
    x = torch.randn(n, h, device=dev) * scale_x
    mem = torch.randn(n, h, device=dev) if mem_kind == "randn" else torch.zeros(n, h, device=dev)
    with torch.no_grad():
        ms = timed(lambda: ops.gru_cell(x, w_ih, w_hh, b, b, mem))
    print(f"{label}: {ms:.3f} ms", flush=True)
# fresh memory every call (the state does not converge)
x = torch.randn(n, h, device=dev)
mems = [torch.randn(n, h, device=dev) for _ in range(12)]
it = iter(range(100))
with torch.no_grad():
    ms = timed(lambda: ops.gru_cell(x, w_ih, w_hh, b, b, mems[next(it) % 12]))
print(f"randn x, a different randn mem each call: {ms:.3f} ms")

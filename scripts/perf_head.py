"""Timing of the one-kernel inference head at the benchmark shape: python scripts/perf_head.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
m = 1_000_000
for i, dims in ((128, [10, 20, 10, 5, 1]), (256, [10, 20, 10, 5, 1]), (64, [32, 32, 1]), (96, [10, 20, 10, 5, 1]), (512, [10, 20, 10, 5, 1]), (384, [10, 20, 10, 5, 1])):
    x = torch.randn(m, i, device=dev)
    layers, w = [], i
    for j, d in enumerate(dims):
        layers.append((torch.randn(d, w, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1, j + 1 < len(dims)))
        w = d
    packed = ops.pack_mlp_head(layers, dev)
    fn = lambda: ops.mlp_head(x, packed=packed)
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    ms = ts[len(ts) // 2]
    print(f"I={i} dims={dims}: {ms:.3f} ms  ({m * (i + dims[-1]) * 4 / ms / 1e9:.2f} TB/s)", flush=True)

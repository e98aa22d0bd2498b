"""Do the snapshot build (radix sorts, HBM-bound) and the recurrent layer (phase-bound, one
workgroup per CU) overlap when they are queued on two streams?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from spiral_amd import graph as G, ops  # noqa: E402

dev = torch.device("cuda:0")
n, e, h, g = 1_000_000, 10_000_000, 128, 3
perm = torch.randperm(n, device=dev)
src, dst = bench.gen_snapshot(n, e, 0, 3.0, 1.5, perm, dev)
x = torch.randn(n, h, device=dev)
w1, w2 = torch.randn(g * h, h, device=dev) * 0.05, torch.randn(g * h, h, device=dev) * 0.05
b = torch.randn(g * h, device=dev) * 0.05
mem = torch.randn(2, n, h, device=dev)
side = torch.cuda.Stream()


def build():
    gr = G.SnapshotGraph(src, dst, n)
    gr.csr(wait=False)
    gr.norm("in", 0)
    gr.norm("out", 0)
    return gr


def cells():
    h1 = ops.gru_cell(x, w1, w2, b, b, mem[0])
    return ops.gru_cell(h1, w1, w2, b, b, mem[1])


def wall(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        keep = build()
    out = cells()
    torch.cuda.current_stream().wait_stream(side)
    return keep, out


with torch.no_grad():
    tb, tc, tboth = wall(build), wall(cells), wall(both)
print(f"build {tb:.3f} ms, two recurrent layers {tc:.3f} ms, sum {tb + tc:.3f} ms, on two streams {tboth:.3f} ms")

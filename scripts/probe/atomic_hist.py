"""How fast are global atomics for an out-degree histogram?  torch.bincount / scatter_add_ on
the benchmark's source-id distribution against the keys-only sort of spmi_degree."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from spiral_amd import graph as G  # noqa: E402

dev = torch.device("cuda:0")
n, e = 1_000_000, 10_000_000
perm = torch.randperm(n, device=dev)
src, dst = bench.gen_snapshot(n, e, 0, 3.0, 1.5, perm, dev)


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for name, ids in (("src", src), ("dst", dst)):
    ones = torch.ones_like(ids, dtype=torch.int32)
    print(name, "max degree", int(torch.bincount(ids, minlength=n).max()))
    print(name, "bincount      ", timed(lambda: torch.bincount(ids, minlength=n)), "ms")
    print(name, "scatter_add   ", timed(lambda: torch.zeros(n, dtype=torch.int32, device=dev).scatter_add_(0, ids, ones)), "ms")
    print(name, "spmi_degree   ", timed(lambda: G.degree(ids, n, validated=True)), "ms")

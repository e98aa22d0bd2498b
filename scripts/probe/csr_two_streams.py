"""Probe: the by-destination CSR build and the out-degree normaliser of one snapshot on two
streams against one after the other (1M nodes, 10M edges)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spiral_amd import graph as G  # noqa: E402

dev = torch.device("cuda:0")
n, e = 1_000_000, 10_000_000
gen = torch.Generator(device=dev)
gen.manual_seed(1)
perm = torch.randperm(n, generator=gen, device=dev)
dst = perm[(n * torch.rand(e, generator=gen, device=dev) ** 3.0).long().clamp_(max=n - 1)]
src = perm[(n * torch.rand(e, generator=gen, device=dev) ** 1.5).long().clamp_(max=n - 1)]
side = torch.cuda.Stream()
torch.set_grad_enabled(False)  # as in inference: the out normaliser then counts ids, it does not build a second CSR


def serial():
    g = G.SnapshotGraph(src, dst, n)
    g.csr(wait=False)
    g.norm("in", 0)
    return g, g.norm("out", 0)          # counts the sorted build's int32 source column


def serial_ids():
    g = G.SnapshotGraph(src, dst, n)
    g.csr(wait=False)
    g.norm("in", 0)
    return g, G.norm_from_degree(G.degree(src, n), 0)   # counts the int64 ids: independent of the build


def forked():
    g = G.SnapshotGraph(src, dst, n)
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = G.norm_from_degree(G.degree(src, n), 0)
    g.csr(wait=False)
    g.norm("in", 0)
    cur.wait_stream(side)
    out.record_stream(cur)
    return g, out


for name, fn in (("one stream, int32 column", serial), ("one stream, int64 ids", serial_ids), ("two streams", forked),
                 ("one stream, int32 column", serial), ("one stream, int64 ids", serial_ids), ("two streams", forked)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"{name}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms", flush=True)
(g1, o1), (g2, o2) = serial(), forked()
torch.cuda.synchronize()
assert torch.equal(o1, o2) and torch.equal(g1.csr().indices, g2.csr().indices)
print("same results")

"""Times the snapshot build (COO -> CSR + row normaliser + hub-row plan, then the source
normaliser) at the headline shape.  Usage: python scripts/perf_csr.py [N E [k_dst]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import graph as G  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
e = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
k_dst = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(99)
perm = torch.randperm(n, generator=gen, device=dev)
gen.manual_seed(1234)
dst = perm[(torch.rand(e, generator=gen, device=dev, dtype=torch.float64).pow_(k_dst) * n).long().clamp_(max=n - 1)]
src = perm[(torch.rand(e, generator=gen, device=dev, dtype=torch.float64).pow_(1.5) * n).long().clamp_(max=n - 1)]


def build():
    g = G.SnapshotGraph(src, dst, n)
    g.csr(wait=False)
    g.norm("in", 0)
    g.norm("out", 0)
    return g


with torch.no_grad():
    for _ in range(3):
        build()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for i in range(20):
        build()
        ev[i + 1].record()
    torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
print(f"N={n} E={e} k_dst={k_dst}: snapshot build median {ms[10]:.3f} ms  min {ms[0]:.3f}  max {ms[-1]:.3f}")

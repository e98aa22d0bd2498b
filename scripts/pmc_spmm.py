"""Workload for the HBM-traffic counters of the fused aggregation (run under rocprofv3 --pmc).

Two launches per repetition:
  1. calibration: a permutation graph (every row has exactly one neighbour, every source
     row is read exactly once), whose compulsory traffic is known:
     reads N*F*4 + N*4 (indices) + (N+1)*4 (row pointers) + N*4 (destination scale) bytes,
     writes N*F*4 bytes;
  2. the benchmark snapshot (power law, 1M nodes x 10M edges, F=128) as the inference path
     launches it: no per-edge weight (the source normaliser was applied by the projection),
     destination scale, bias and leaky_relu fused.
The kernel names are identical, so launches are told apart by order (odd = calibration).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
from spiral_amd import graph as G  # noqa: E402
from perf_spmm import powerlaw  # noqa: E402

# usage: pmc_spmm.py [F] [k_dst] [N] [E]   (defaults 128, 3.0, 1M, 10M: the headline workload)
f = int(sys.argv[1]) if len(sys.argv) > 1 else 128
k_dst = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
e = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
dev = torch.device("cuda:0")
n_cal = 10_000_000 if f <= 128 else 5_000_000
perm = torch.randperm(n_cal, device=dev)
rows = torch.arange(n_cal, device=dev)
csr_cal = G.build_csr(rows, perm, n_cal, n_cal)
x_cal = torch.randn(n_cal, f, device=dev)
out_cal = torch.empty(n_cal, f, device=dev)
ew_cal = torch.ones(n_cal, device=dev)
ds_cal = torch.ones(n_cal, device=dev)

src, dst = powerlaw(n, e, 0, k_dst, 1.5, dev)
csr = G.build_csr(dst, src, n, n)
csr_t = G.build_csr(src, dst, n, n)
x = torch.randn(n, f, device=dev)
ew = csr_t.norm(0).index_select(0, csr.indices)
ds = csr.norm(0)
bias = torch.zeros(f, device=dev)
out = torch.empty(n, f, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    G.aggregate(csr_cal, x_cal, None, None, ds_cal, None, None, out=out_cal)
    G.aggregate(csr, x, None, None, ds, bias, "leaky_relu", 0.01, out=out)
torch.cuda.synchronize()
print("done")

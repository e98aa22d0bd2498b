import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spiral_amd import graph as G
dev = torch.device("cuda:0")
n, e = 1_000_000, 10_000_000
perm = torch.randperm(n, device=dev)
for k_src in (1.5, 3.0, 8.0):
    src, dst = bench.gen_snapshot(n, e, 0, 3.0, k_src, perm, dev)
    ids = src.to(torch.int32).contiguous()
    ref = torch.bincount(src, minlength=n).clamp(min=1).float().pow(-0.5)
    got = G.norm_from_ids32(ids, n, 0)
    ok = torch.equal(got, (1.0 / torch.bincount(src, minlength=n).clamp(min=1).float().sqrt()))
    close = float((got - ref).abs().max())
    evs = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); G.norm_from_ids32(ids, n, 0); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"k_src {k_src}: max multiplicity {int(torch.bincount(src).max())}, exact {ok} maxdiff {close:.2e}, median {ts[10]*1e3:.1f} us best {ts[0]*1e3:.1f} us", flush=True)

"""What one rank's per-snapshot bookkeeping costs on the device, without a second GPU: the exchange
plan, the out-degree exchange and the CSR build of ONE rank of a P-rank partition of a workload,
with the collectives replaced by local stand-ins that return data of the right shape (every peer
is assumed to want from this rank what this rank wants from it).  The transfers themselves are
not what is measured; everything around them is: marks and scan over the node range, the one
host round trip, compaction, table numbering, partial out-degrees, the sort.

    python scripts/perf_plan.py [workload] [P ...]        e.g.  config3 2 4 8
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import graph as G  # noqa: E402
from spiral_amd import parallel as par  # noqa: E402

SIZES = {"config2": (1_000_000, 10_000_000, 3.0), "config3": (10_000_000, 100_000_000, 3.0),
         "config5": (1_000_000, 10_000_000, 6.0)}
workload = sys.argv[1] if len(sys.argv) > 1 else "config3"
ranks = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
n, e, k_dst = SIZES[workload]
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(99)
perm = torch.randperm(n, generator=gen, device=dev)
gen.manual_seed(1234)
dst = perm[(torch.rand(e, generator=gen, device=dev, dtype=torch.float64).pow_(k_dst) * n).long().clamp_(max=n - 1)]
src = perm[(torch.rand(e, generator=gen, device=dev, dtype=torch.float64).pow_(1.5) * n).long().clamp_(max=n - 1)]
deg = G.degree(dst, n)

state = {}


def all_gather_into_tensor(out, inp, group=None):
    p = state["world"]
    m = out.view(p, p)
    m.zero_()
    m[state["rank"]] = inp          # what this rank needs from everyone
    m[:, state["rank"]] = inp       # ... and everyone is taken to need the same from it


def all_to_all_single(out, inp, out_splits=None, in_splits=None, group=None, async_op=False):
    if out.dim() == 2 and out.shape[1] == 2:               # requests: (row in the owner's block, count)
        out.copy_(inp)
        out[:, 0].remainder_(state["n_local"])             # a row of THIS rank's block, as a peer would ask
    else:
        out.copy_(inp.reshape(out.shape))


dist.all_gather_into_tensor = all_gather_into_tensor
dist.all_to_all_single = all_to_all_single

for world in ranks:
    bounds = par.edge_balanced_bounds(deg, world)
    rank = world // 2
    part = par.RowPartition(n, bounds, rank, world)
    lo, hi = part.block()
    keep = (dst >= lo) & (dst < hi)
    s, d = src[keep].contiguous(), dst[keep].contiguous()
    state.update(world=world, rank=rank, n_local=hi - lo)
    times = {"plan": [], "csr_build": [], "out_degree": []}
    for it in range(6):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        shard = par.ShardedSnapshot(s, d, part)
        ev[1].record()
        shard.batch_csrs()
        shard.norm("in", 0)
        ev[2].record()
        shard.norm("out", 0)
        ev[3].record()
        torch.cuda.synchronize()
        if it >= 2:
            for i, k in enumerate(times):
                times[k].append(ev[i].elapsed_time(ev[i + 1]))
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    print(f"{workload} P={world} rank {rank}: rows {hi - lo}, edges {s.numel()}, remote rows named "
          f"{shard.plan.n_remote} of {n - (hi - lo)}; plan {med['plan']:.3f} ms, CSR build {med['csr_build']:.3f} ms, "
          f"out-degree exchange (local part) {med['out_degree']:.3f} ms", flush=True)
    del shard, s, d, keep

if os.environ.get("PLAN_PROFILE") == "1":   # the kernels of one plan, longest first
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        bounds = par.edge_balanced_bounds(deg, ranks[-1])
        part = par.RowPartition(n, bounds, ranks[-1] // 2, ranks[-1])
        lo, hi = part.block()
        keep = (dst >= lo) & (dst < hi)
        s, d = src[keep].contiguous(), dst[keep].contiguous()
        state.update(world=ranks[-1], rank=ranks[-1] // 2, n_local=hi - lo)
        par.ShardedSnapshot(s, d, part)     # workspace made here
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        shard = par.ShardedSnapshot(s, d, part)
        torch.cuda.synchronize()
        print(f"second plan, wall {1e3 * (time.perf_counter() - t0):.3f} ms")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))

"""The row-partitioned GraphConv with the real kernels: two ranks share the one GPU of the
test box and exchange through gloo with host staging (RCCL needs a device per rank, so
the collectives themselves are not under test here, only everything around them).
Sharded forward and backward must equal the single-GPU layer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _stage_through_host():
    class Done:
        def wait(self):
            return True

    def staged(fn):
        def call(*args, **kw):
            args = list(args)
            host = [a.cpu() if isinstance(a, torch.Tensor) else a for a in args]
            was_async = kw.pop("async_op", False)
            if host[0].dtype == torch.bfloat16:  # gloo moves bytes
                host = [a.view(torch.uint8) if isinstance(a, torch.Tensor) else a for a in host]
            r = fn(*host, **kw)
            args[0].copy_(host[0].view(args[0].dtype))
            return Done() if was_async else r
        return call

    for name in ("all_gather_into_tensor", "reduce_scatter_tensor", "all_reduce"):
        setattr(dist, name, staged(getattr(dist, name)))


def _problem():
    from oracle import oracle as orc

    n, e, fi, fo = 3000, 40_000, 128, 128
    rng = np.random.default_rng(0)
    src, dst = orc.powerlaw_snapshot(n, e, t=2, k_dst=5.0)
    x = rng.standard_normal((n, fi)).astype(np.float32)
    w = (rng.standard_normal((fi, fo)) * 0.1).astype(np.float32)
    b = rng.standard_normal(fo).astype(np.float32)
    return n, src, dst, x, w, b


def _layer(fi, fo, w, b, dev):
    import torch.nn.functional as F
    from spiral_amd.core.types import GraphConv

    conv = GraphConv(fi, fo, activation=F.leaky_relu, allow_zero_in_degree=True).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
        conv.bias.copy_(torch.from_numpy(b))
    return conv


def _two_layers(fi, dev):
    import torch.nn.functional as F
    from spiral_amd.core.types import GraphConv

    torch.manual_seed(5)
    conv1 = GraphConv(fi, 64, activation=F.leaky_relu, allow_zero_in_degree=True).to(dev)
    conv2 = GraphConv(64, 96, activation=F.relu, allow_zero_in_degree=True).to(dev)  # aggregates first
    with torch.no_grad():
        conv1.bias.normal_()
        conv2.bias.normal_()
    return conv1, conv2


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spiral_amd import graph as G
        from spiral_amd import parallel as par

        _stage_through_host()
        dev = torch.device("cuda:0")
        n, src, dst, x, w, b = _problem()
        src_t, dst_t = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
        bounds = par.edge_balanced_bounds(G.degree(dst_t, n), world)
        part = par.RowPartition(n, bounds, rank, world)
        lo, hi = part.block()
        shard = par.ShardedSnapshot.from_global(src_t, dst_t, part)
        conv = _layer(x.shape[1], w.shape[1], w, b, dev)
        xl = torch.tensor(x[lo:hi], device=dev, requires_grad=True)
        y = par.sharded_graphconv(conv, shard, xl)
        (y ** 2).sum().backward()
        par.allreduce_gradients(conv)
        out = dict(bounds=bounds, y=y.detach().cpu().numpy(), gx=xl.grad.cpu().numpy(),
                   gw=conv.weight.grad.cpu().numpy(), gb=conv.bias.grad.cpu().numpy())

        # the layer in two halves: replicated input (no exchange) for layer 1, the gather of
        # layer 2 in flight while the next snapshot's front half runs
        from oracle import oracle as orc

        conv1, conv2 = _two_layers(x.shape[1], dev)
        x_repl = par.replicate_rows(torch.from_numpy(x).to(dev), part)
        deg_group = dist.new_group()
        in_flight, h2, h2_bf16 = [], [], []
        with torch.no_grad():
            for t in range(3):
                s_np, d_np = orc.powerlaw_snapshot(n, 30_000, t=t, k_dst=4.0)
                sh = par.ShardedSnapshot.from_global(torch.from_numpy(s_np).to(dev),
                                                     torch.from_numpy(d_np).to(dev), part, deg_group)
                h1 = par.graphconv_replicated(conv1, sh, x_repl)   # aggregate into the block, then project
                z2 = par.graphconv_project(conv2, h1)
                in_flight.append((sh, par.gather_rows_async(z2, part),
                                  par.gather_rows_async(z2, part, None, torch.bfloat16)))
                if len(in_flight) > 1:
                    sh0, p32, p16 = in_flight.pop(0)
                    h2.append(par.graphconv_aggregate(conv2, sh0, p32.wait()).cpu().numpy())
                    h2_bf16.append(par.graphconv_aggregate(conv2, sh0, p16.wait()).cpu().numpy())
            while in_flight:
                sh0, p32, p16 = in_flight.pop(0)
                h2.append(par.graphconv_aggregate(conv2, sh0, p32.wait()).cpu().numpy())
                h2_bf16.append(par.graphconv_aggregate(conv2, sh0, p16.wait()).cpu().numpy())
        out.update(h2=h2, h2_bf16=h2_bf16)

        # gradients through the replicated-input layer: every rank's partial, summed
        conv1.zero_grad()
        y1 = par.graphconv_aggregate(conv1, shard, par.graphconv_project(conv1, x_repl))
        (y1 ** 2).sum().backward()
        par.allreduce_gradients(conv1)
        out.update(y1=y1.detach().cpu().numpy(), gw1=conv1.weight.grad.cpu().numpy(),
                   gb1=conv1.bias.grad.cpu().numpy())
        # the same layer in the order that projects only the local block: with and without autograd
        conv1.zero_grad()
        y1r = par.graphconv_replicated(conv1, shard, x_repl)
        (y1r ** 2).sum().backward()
        par.allreduce_gradients(conv1)
        with torch.no_grad():
            y1r_ng = par.graphconv_replicated(conv1, shard, x_repl)
        out.update(y1r=y1r.detach().cpu().numpy(), y1r_ng=y1r_ng.cpu().numpy(),
                   gw1r=conv1.weight.grad.cpu().numpy())
        results[rank] = out
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_single_gpu(dev):
    from spiral_amd.graph import SnapshotGraph

    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
        results = dict(results)
    n, src, dst, x, w, b = _problem()
    g = SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
    conv = _layer(x.shape[1], w.shape[1], w, b, dev)
    xg = torch.tensor(x, device=dev, requires_grad=True)
    y = conv(g, xg)
    (y ** 2).sum().backward()

    def rel(a, ref):
        ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else ref
        return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)

    bounds = results[0]["bounds"]
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        assert rel(results[r]["y"], y[lo:hi]) < 1e-5
        assert rel(results[r]["gx"], xg.grad[lo:hi]) < 1e-5
        assert rel(results[r]["gw"], conv.weight.grad) < 1e-5
        assert rel(results[r]["gb"], conv.bias.grad) < 1e-5


    # two-half layers, exchange in flight across snapshots: same numbers as the plain order
    from oracle import oracle as orc

    conv1, conv2 = _two_layers(x.shape[1], dev)
    xt = torch.from_numpy(x).to(dev)
    for t in range(3):
        s_np, d_np = orc.powerlaw_snapshot(n, 30_000, t=t, k_dst=4.0)
        gt = SnapshotGraph(torch.from_numpy(s_np).to(dev), torch.from_numpy(d_np).to(dev), n)
        with torch.no_grad():
            want = conv2(gt, conv1(gt, xt))
        for r in range(world):
            lo, hi = bounds[r], bounds[r + 1]
            assert rel(results[r]["h2"][t], want[lo:hi]) < 1e-5
            # bfloat16 payload: inputs rounded once (2^-9 relative), float32 sums
            assert rel(results[r]["h2_bf16"][t], want[lo:hi]) < 1e-2
    conv1.zero_grad()
    y1 = conv1(g, xt)
    (y1 ** 2).sum().backward()
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        assert rel(results[r]["y1"], y1[lo:hi]) < 1e-5
        assert rel(results[r]["gw1"], conv1.weight.grad) < 1e-5
        assert rel(results[r]["y1r"], y1[lo:hi]) < 1e-5
        assert rel(results[r]["y1r_ng"], y1[lo:hi]) < 1e-5
        assert rel(results[r]["gw1r"], conv1.weight.grad) < 1e-5
        assert rel(results[r]["gb1"], conv1.bias.grad) < 1e-5

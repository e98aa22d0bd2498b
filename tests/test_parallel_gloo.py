"""The exchange logic of the row-partitioned path with two gloo processes on the CPU.

No GPU here, so the local kernels are replaced by the CPU oracle (as a stand-in for
libspmi; the kernels themselves are checked in the -m gpu tests).  What is under test:
edge-balanced block bounds, the padded all-gather layout and id remapping, the
reduce-scatter of the transposed aggregation, the out-degree all-reduce and the
gradient all-reduce.  The sharded result must equal the single-process oracle."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc
from spiral_amd import parallel as par


def _cpu_kernels():
    class CpuCSR:
        def __init__(self, indptr, indices, eid, n_rows, n_cols):
            self.indptr, self.indices, self.eid = (torch.from_numpy(a) for a in (indptr, indices, eid))
            self.n_rows, self.n_cols, self.nnz = n_rows, n_cols, len(indices)

        def degrees(self):
            return (self.indptr[1:] - self.indptr[:-1]).to(torch.int64)

    def build_csr(key, other, n_key, n_other, *_):
        return CpuCSR(*orc.coo_to_csr(key.numpy(), other.numpy(), n_key, n_other), n_key, n_other)

    def aggregate(csr, x, src_scale=None, edge_w=None, dst_scale=None, bias=None, act=None, slope=0.01):
        n = lambda t: None if t is None else t.detach().numpy()
        assert x.shape[0] == csr.n_cols
        return torch.from_numpy(orc.aggregate(csr.indptr.numpy(), csr.indices.numpy(), n(x), n(src_scale), n(edge_w),
                                              n(dst_scale), n(bias), act, slope))

    def degree(ids, n):
        return torch.from_numpy(orc.degree(ids.numpy(), n))

    def act_bias_bwd(g_y, y, act, slope, want_gb):
        g = g_y.clone()
        if act == "relu":
            g = torch.where(y > 0, g, torch.zeros_like(g))
        elif act == "leaky_relu":
            g = torch.where(y > 0, g, g * slope)
        return g, (g.sum(0) if want_gb else None)

    return SimpleNamespace(build_csr=build_csr, aggregate=aggregate, degree=degree,
                           act_bias_bwd=act_bias_bwd)


def _problem(fi, fo):
    n, e = 600, 6000
    rng = np.random.default_rng(fi + fo)
    src, dst = orc.powerlaw_snapshot(n, e, t=1, k_dst=4.0)
    x = rng.standard_normal((n, fi)).astype(np.float32)
    w = (rng.standard_normal((fi, fo)) * 0.3).astype(np.float32)
    b = rng.standard_normal(fo).astype(np.float32)
    return n, src, dst, x, w, b


def _worker(rank, world, port, fi, fo, act, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        from spiral_amd.core.types import GraphConv

        par.kernels = _cpu_kernels()
        n, src, dst, x, w, b = _problem(fi, fo)
        bounds = par.edge_balanced_bounds(torch.from_numpy(orc.degree(dst, n)), world)
        part = par.RowPartition(n, bounds, rank, world)
        lo, hi = part.block()
        shard = par.ShardedSnapshot.from_global(torch.from_numpy(src), torch.from_numpy(dst), part)
        conv = GraphConv(fi, fo, activation={"relu": F.relu, None: None}[act], allow_zero_in_degree=True)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w))
            conv.bias.copy_(torch.from_numpy(b))
        xl = torch.tensor(x[lo:hi], requires_grad=True)
        y = par.sharded_graphconv(conv, shard, xl)
        (y ** 2).sum().backward()
        par.allreduce_gradients(conv)
        results[rank] = dict(bounds=bounds, y=y.detach().numpy(), gx=xl.grad.numpy(),
                             gw=conv.weight.grad.numpy(), gb=conv.bias.grad.numpy(),
                             nnz=shard.csr().nnz)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("fi,fo,act", [(12, 6, "relu"), (8, 8, None)])
def test_two_rank_graphconv_equals_single_process(fi, fo, act):
    world = 2
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), fi, fo, act, results), nprocs=world, join=True)
        results = dict(results)
    n, src, dst, x, w, b = _problem(fi, fo)
    xt, wt, bt = (torch.tensor(a, requires_grad=True) for a in (x, w, b))
    y_ref = orc.graphconv_torch(src, dst, n, xt, wt, bt, act=act)
    (y_ref ** 2).sum().backward()

    def rel(a, ref):
        return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)

    bounds = results[0]["bounds"]
    assert bounds == results[1]["bounds"] and bounds[0] == 0 and bounds[-1] == n
    e = len(src)
    assert results[0]["nnz"] + results[1]["nnz"] == e
    assert abs(results[0]["nnz"] - e / 2) < 0.15 * e  # cut at equal nonzeros, hub rows permitting
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        assert rel(results[r]["y"], y_ref.detach().numpy()[lo:hi]) < 1e-5
        assert rel(results[r]["gx"], xt.grad.numpy()[lo:hi]) < 1e-5
        assert rel(results[r]["gw"], wt.grad.numpy()) < 1e-5
        assert rel(results[r]["gb"], bt.grad.numpy()) < 1e-5


def test_bounds_and_padded_layout():
    deg = torch.tensor([5, 0, 0, 1, 1, 1, 1, 1, 0, 0])
    b = par.edge_balanced_bounds(deg, 2)
    assert b == [0, 1, 10] or b == [0, 3, 10] or b[1] in (1, 2, 3, 4)
    assert sum(deg[: b[1]]) >= 5
    part = par.RowPartition(10, [0, 3, 10], 1, 2)
    assert part.max_block == 7 and part.n_gather == 14 and part.block() == (3, 10)
    ids = torch.tensor([0, 2, 3, 9])
    assert part.padded_rows(ids).tolist() == [0, 2, 7, 13]
    one = par.edge_balanced_bounds(deg, 1)
    assert one == [0, 10]


def _gather_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, f = 11, 3
        part = par.RowPartition(n, [0, 4, 11], rank, world)
        x = torch.arange(n * f, dtype=torch.float32).reshape(n, f)
        lo, hi = part.block()
        pending = par.gather_rows_async(x[lo:hi] * 2.0, part)
        repl = par.replicate_rows(x, part)  # independent work while the rows travel
        results[rank] = dict(gathered=pending.wait().numpy(), repl=repl.numpy())
    finally:
        dist.destroy_process_group()


def test_async_gather_and_replicated_layout_agree():
    """The gathered matrix of per-rank blocks and the replicated layout of the same global
    matrix address rows identically (RowPartition.padded_rows); padding rows are zero."""
    world = 2
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_gather_worker, args=(world, _free_port(), results), nprocs=world, join=True)
        results = dict(results)
    n, f = 11, 3
    x = np.arange(n * f, dtype=np.float32).reshape(n, f)
    part = par.RowPartition(n, [0, 4, 11], 0, world)
    rows = part.padded_rows(torch.arange(n)).numpy()
    for r in range(world):
        assert results[r]["gathered"].shape == (part.n_gather, f)
        np.testing.assert_array_equal(results[r]["gathered"][rows], 2.0 * x)
        np.testing.assert_array_equal(results[r]["repl"][rows], x)
        pad = np.setdiff1d(np.arange(part.n_gather), rows)
        assert not results[r]["repl"][pad].any() and not results[r]["gathered"][pad].any()

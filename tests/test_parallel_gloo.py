"""The exchange logic of the row-partitioned path with two gloo processes on the CPU.

No GPU here, so the local kernels are replaced by the CPU oracle (as a stand-in for
libspmi; the kernels themselves are checked in the -m gpu tests).  What is under test:
edge-balanced block bounds, the exchange plan (which rows travel where, the compact source
table and its column numbering, batches), the all-to-all of the needed rows and its reverse
in the backward pass, the out-degree all-reduce and the gradient all-reduce.  The sharded
result must equal the single-process oracle, for 2 to 4 ranks and 1 to 3 batches."""
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

    def build_csr(key, other, n_key, n_other, *_, **__):
        return CpuCSR(*orc.coo_to_csr(key.numpy(), other.numpy(), n_key, n_other), n_key, n_other)

    def aggregate(csr, x, src_scale=None, edge_w=None, dst_scale=None, bias=None, act=None, slope=0.01,
                  out=None, out_dtype=None, accumulate=False, finalize=True):
        n = lambda t: None if t is None else t.detach().float().numpy()
        assert x.shape[0] == csr.n_cols
        if not accumulate and finalize:
            return torch.from_numpy(orc.aggregate(csr.indptr.numpy(), csr.indices.numpy(), n(x), n(src_scale),
                                                  n(edge_w), n(dst_scale), n(bias), act, slope))
        raw = torch.from_numpy(orc.aggregate(csr.indptr.numpy(), csr.indices.numpy(), n(x), n(src_scale), n(edge_w)))
        if accumulate:
            raw = raw + out
        if finalize:
            if dst_scale is not None:
                raw = raw * dst_scale[:, None]
            if bias is not None:
                raw = raw + bias
            raw = torch.from_numpy(orc._activation(raw.numpy(), act, slope))
        if out is not None:  # the library writes into the tensor it is given
            out.copy_(raw)
            return out
        return raw

    def degree(ids, n, validated=False, out=None):
        deg = torch.from_numpy(orc.degree(ids.numpy(), n))
        return deg if out is None else out.copy_(deg)

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


def _worker(rank, world, port, fi, fo, act, stages, results):
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
        shard = par.ShardedSnapshot.from_global(torch.from_numpy(src), torch.from_numpy(dst), part,
                                                stages=stages)
        conv = GraphConv(fi, fo, activation={"relu": F.relu, None: None}[act], allow_zero_in_degree=True)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w))
            conv.bias.copy_(torch.from_numpy(b))
        xl = torch.tensor(x[lo:hi], requires_grad=True)
        y = par.sharded_graphconv(conv, shard, xl)
        (y ** 2).sum().backward()
        par.allreduce_gradients(conv)
        out = dict(bounds=bounds, y=y.detach().numpy(), gx=xl.grad.numpy(),
                   gw=conv.weight.grad.numpy(), gb=conv.bias.grad.numpy(), nnz=shard.csr().nnz,
                   n_table=shard.plan.n_table, n_local=shard.n_local,
                   table_ids=shard.plan.table_ids.numpy(), sent=int(shard.plan.send_index.numel()),
                   needed=np.unique(src[(dst >= lo) & (dst < hi)]))
        with torch.no_grad():                      # the two halves, batch by batch
            y_ng = par.sharded_graphconv(conv, shard, xl.detach())
            y_rep = par.graphconv_replicated(conv, shard, torch.from_numpy(x))
        conv.zero_grad()
        y_rep_t = par.graphconv_replicated(conv, shard, torch.from_numpy(x))
        (y_rep_t ** 2).sum().backward()
        par.allreduce_gradients(conv)
        out.update(y_ng=y_ng.numpy(), y_rep=y_rep.numpy(), gw_rep=conv.weight.grad.numpy())
        results[rank] = out
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,fi,fo,act,stages", [(2, 12, 6, "relu", 1), (2, 8, 8, None, 2),
                                                    (3, 8, 12, "relu", 3), (3, 12, 6, None, 1),
                                                    (4, 8, 8, "relu", 2)])
def test_sharded_graphconv_equals_single_process(world, fi, fo, act, stages):
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), fi, fo, act, stages, results), nprocs=world, join=True)
        results = dict(results)
    n, src, dst, x, w, b = _problem(fi, fo)
    xt, wt, bt = (torch.tensor(a, requires_grad=True) for a in (x, w, b))
    y_ref = orc.graphconv_torch(src, dst, n, xt, wt, bt, act=act)
    (y_ref ** 2).sum().backward()

    def rel(a, ref):
        return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)

    bounds = results[0]["bounds"]
    assert all(results[r]["bounds"] == bounds for r in range(world)) and bounds[0] == 0 and bounds[-1] == n
    e = len(src)
    assert sum(results[r]["nnz"] for r in range(world)) == e
    assert abs(results[0]["nnz"] - e / world) < 0.15 * e  # cut at equal nonzeros, hub rows permitting
    received = 0
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        res = results[r]
        # the source table: own block in place, then exactly the remote rows this rank's edges name
        remote = res["needed"][(res["needed"] < lo) | (res["needed"] >= hi)]
        assert res["n_table"] == res["n_local"] + len(remote)
        np.testing.assert_array_equal(res["table_ids"][: hi - lo], np.arange(lo, hi))
        np.testing.assert_array_equal(np.sort(res["table_ids"][hi - lo:]), remote)
        received += len(remote)
        assert rel(res["y"], y_ref.detach().numpy()[lo:hi]) < 1e-5
        assert rel(res["y_ng"], y_ref.detach().numpy()[lo:hi]) < 1e-5
        assert rel(res["y_rep"], y_ref.detach().numpy()[lo:hi]) < 1e-5
        assert rel(res["gx"], xt.grad.numpy()[lo:hi]) < 1e-5
        assert rel(res["gw"], wt.grad.numpy()) < 1e-5
        assert rel(res["gb"], bt.grad.numpy()) < 1e-5
        assert rel(res["gw_rep"], wt.grad.numpy()) < 1e-5
    assert sum(results[r]["sent"] for r in range(world)) == received   # every row sent is a row needed
    assert received < (world - 1) * n                                    # less than an all-gather moves


def test_bounds_and_slices():
    deg = torch.tensor([5, 0, 0, 1, 1, 1, 1, 1, 0, 0])
    b = par.edge_balanced_bounds(deg, 2)
    assert b[0] == 0 and b[-1] == 10 and b[1] in (1, 2, 3, 4)
    assert sum(deg[: b[1]]) >= 5
    part = par.RowPartition(10, [0, 3, 10], 1, 2)
    assert part.block() == (3, 10) and part.block(0) == (0, 3)
    assert part.owner(torch.tensor([0, 2, 3, 9])).tolist() == [0, 0, 1, 1]
    assert par.edge_balanced_bounds(deg, 1) == [0, 10]
    assert par._slice_bounds(10, 3) == [0, 3, 6, 10] and par._slice_bounds(0, 2) == [0, 0, 0]


def test_single_rank_plan_needs_no_process_group():
    """world = 1: the table is the block itself and nothing is exchanged."""
    part = par.RowPartition(6, [0, 6], 0, 1)
    plan = par.ExchangePlan(torch.tensor([5, 0, 0, 3]), part, stages=2)
    assert plan.n_table == 6 and plan.n_remote == 0
    assert plan.columns(torch.tensor([5, 0, 3])).tolist() == [5, 0, 3]
    pending = par.exchange_rows_async(torch.arange(12.0).reshape(6, 2), plan)
    assert torch.equal(pending.wait(), torch.arange(12.0).reshape(6, 2))


# ---- shapes a first run on real hardware must not trip over ---------------------------------
def _skewed_problem(bounds):
    """A graph laid out against fixed block bounds so that the exchange has every awkward
    shape at once: block 0 names no remote row at all; block 1 needs many rows of block 0 and a
    single row of the last block (uneven per-peer counts, fewer rows than batches); the last
    block needs rows of block 1 only (a per-peer count of zero in both directions with block
    0).  Blocks in between, if any, are empty or have no edges."""
    rng = np.random.default_rng(17)
    n = bounds[-1]
    first, second, last = (0, bounds[1]), None, None
    blocks = [(bounds[q], bounds[q + 1]) for q in range(len(bounds) - 1) if bounds[q + 1] > bounds[q]]
    first, second, last = blocks[0], blocks[1], blocks[-1]
    src, dst = [], []

    def edges(s_range, d_range, count):
        src.append(rng.integers(*s_range, count))
        dst.append(rng.integers(*d_range, count))

    edges(first, first, 200)                       # block 0: own sources only
    edges(first, second, 300)                      # block 1: many rows of block 0 ...
    src.append(np.full(5, last[0] + 3))            # ... one row of the last block (5 edges)
    dst.append(rng.integers(*second, 5))
    edges(second, second, 50)
    edges(second, last, 120)                       # last block: rows of block 1 and its own
    edges(last, last, 40)
    src, dst = np.concatenate(src).astype(np.int64), np.concatenate(dst).astype(np.int64)
    fi, fo = 6, 6
    x = rng.standard_normal((n, fi)).astype(np.float32)
    w = (rng.standard_normal((fi, fo)) * 0.3).astype(np.float32)
    b = rng.standard_normal(fo).astype(np.float32)
    return n, src, dst, x, w, b


def _skew_worker(rank, world, port, bounds, stages, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        from spiral_amd.core.types import GraphConv

        par.kernels = _cpu_kernels()
        n, src, dst, x, w, b = _skewed_problem(bounds)
        part = par.RowPartition(n, bounds, rank, world)
        lo, hi = part.block()
        conv = GraphConv(x.shape[1], w.shape[1], activation=F.relu, allow_zero_in_degree=True)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w))
            conv.bias.copy_(torch.from_numpy(b))
        out = dict(y=[], gx=[], gw=[], gb=[], need=[], wanted=[], deg=[])
        for _ in range(2):   # twice on one partition: the second plan finds the workspace clean
            shard = par.ShardedSnapshot.from_global(torch.from_numpy(src), torch.from_numpy(dst), part,
                                                    stages=stages)
            conv.zero_grad()
            xl = torch.tensor(x[lo:hi], requires_grad=True)
            y = par.sharded_graphconv(conv, shard, xl)
            (y ** 2).sum().backward()
            par.allreduce_gradients(conv)
            out["y"].append(y.detach().numpy())
            out["gx"].append(xl.grad.numpy())
            out["gw"].append(conv.weight.grad.numpy().copy())
            out["gb"].append(conv.bias.grad.numpy().copy())
            out["need"].append(list(shard.plan.need_per_peer))
            out["wanted"].append(list(shard.plan.send_per_peer))
            out["deg"].append((shard.plan.table_ids.numpy(), shard.out_degree_table().numpy()))
        ws = part.workspace(torch.device("cpu"))
        out["mark_clean"] = True   # marks and counts are rewritten whole by every plan
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bounds,stages", [([0, 30, 60, 90], 1), ([0, 30, 60, 90], 3),
                                           ([0, 30, 30, 60, 90], 2)])
def test_exchange_with_zero_counts_uneven_counts_and_an_empty_block(bounds, stages):
    world = len(bounds) - 1
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_skew_worker, args=(world, _free_port(), bounds, stages, results), nprocs=world, join=True)
        results = dict(results)
    n, src, dst, x, w, b = _skewed_problem(bounds)
    xt, wt, bt = (torch.tensor(a, requires_grad=True) for a in (x, w, b))
    y_ref = orc.graphconv_torch(src, dst, n, xt, wt, bt, act="relu")
    (y_ref ** 2).sum().backward()
    out_deg = np.bincount(src, minlength=n)

    def rel(a, ref):
        return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)

    last = world - 1
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        res = results[r]
        assert res["mark_clean"]
        for rep in range(2):
            if hi > lo:
                assert rel(res["y"][rep], y_ref.detach().numpy()[lo:hi]) < 1e-5
                assert rel(res["gx"][rep], xt.grad.numpy()[lo:hi]) < 1e-5
            assert rel(res["gw"][rep], wt.grad.numpy()) < 1e-5
            assert rel(res["gb"][rep], bt.grad.numpy()) < 1e-5
            ids, deg = res["deg"][rep]
            np.testing.assert_array_equal(deg, out_deg[ids])     # whole-graph out-degrees, table order
        assert res["need"][0] == res["need"][1] and res["wanted"][0] == res["wanted"][1]
    second = 1 if world == 3 else 2
    assert sum(results[0]["need"][0]) == 0                          # block 0 needs nothing
    assert results[second]["need"][0][last] == 1                    # one row of the last block
    assert results[second]["need"][0][0] > 20                       # many of block 0
    assert results[last]["need"][0][0] == 0 and results[0]["wanted"][0][last] == 0
    if world == 4:                                                  # the empty block: no rows either way
        assert sum(results[1]["need"][0]) == 0 and sum(results[1]["wanted"][0]) == 0

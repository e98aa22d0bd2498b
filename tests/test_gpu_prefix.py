"""GPU parity of the cumulative container: the CSR of a time prefix cut from the table-wide
grouping must equal, bit for bit, the CSR the reference's way produces (a fresh grouping of
`edges[:anchor[t]]`, spiro/dtdg/types.py:183-198), checked against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from spiral_amd import _native as nv
from spiral_amd import graph as G
from spiral_amd.dtdg.types import CitationGraph, VEInvariantDTDG

pytestmark = pytest.mark.gpu


def _check_prefixes(src, dst, n_all, cuts, dev):
    """cuts: iterable of (e_cut, n_rows)."""
    cum = G.CumulativeCSR(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n_all, n_all)
    for e_cut, n_rows in cuts:
        got = cum.prefix(e_cut, n_rows, n_rows, want_eid=True)
        indptr, indices, eid = orc.coo_to_csr(dst[:e_cut], src[:e_cut], n_rows)
        assert np.array_equal(got.indptr.cpu().numpy(), indptr), (e_cut, n_rows)
        assert np.array_equal(got.indices.cpu().numpy(), indices), (e_cut, n_rows)
        assert np.array_equal(got.eid.cpu().numpy(), eid), (e_cut, n_rows)
        lean = cum.prefix(e_cut, n_rows, n_rows)
        assert lean.eid is None
        assert np.array_equal(lean.indices.cpu().numpy(), indices)
        deg = cum.prefix_degree(e_cut, n_rows)
        assert deg.dtype == torch.int64
        assert np.array_equal(deg.cpu().numpy(), orc.degree(dst[:e_cut], n_rows))
    return cum


def _growing_table(n, e, seed, k_dst=3.0):
    """Edges whose endpoints grow with time, so that early prefixes touch only early nodes
    (what a time-sorted citation table looks like)."""
    rng = np.random.default_rng(seed)
    frac = (np.arange(e) + 1) / e
    hi = np.maximum(1, np.ceil(frac * n)).astype(np.int64)
    dst = np.minimum((hi * rng.random(e) ** k_dst).astype(np.int64), hi - 1)
    src = np.minimum((hi * rng.random(e)).astype(np.int64), hi - 1)
    return src, dst, hi


@pytest.mark.parametrize("n,e,seed", [(1, 7, 0), (50, 300, 1), (3000, 40_000, 2), (100_000, 1_000_000, 3)])
def test_prefix_equals_fresh_grouping(dev, n, e, seed):
    src, dst, hi = _growing_table(n, e, seed)
    cuts = [(0, 1), (e, n), (1, int(hi[0]))]
    for frac in (0.1, 0.37, 0.5, 0.93):
        c = max(1, int(e * frac))
        cuts.append((c, int(hi[c - 1])))   # tightest consistent node count
        cuts.append((c, n))                # all rows, later ones empty
    _check_prefixes(src, dst, n, cuts, dev)


def test_prefix_sparse_rows_and_hubs(dev):
    # few edges over many rows: a copy tile spans far more rows than fit in LDS
    n, e = 400_000, 9000
    rng = np.random.default_rng(5)
    src = rng.integers(0, n, e, dtype=np.int64)
    dst = rng.integers(0, n, e, dtype=np.int64)
    _check_prefixes(src, dst, n, [(e, n), (e // 2, n), (1, n), (4097, n)], dev)
    # one hub row holding most of the table, spread over time, plus multi-edges
    n, e = 5000, 300_000
    dst = np.where(rng.random(e) < 0.7, 1234, rng.integers(0, n, e)).astype(np.int64)
    src = rng.integers(0, 50, e, dtype=np.int64)
    _check_prefixes(src, dst, n, [(e, n), (e // 3, n), (2048, n), (2049, n), (100_000, n)], dev)


def test_prefix_reports_edges_ahead_of_their_nodes(dev):
    src = np.array([0, 1, 2, 7, 3], dtype=np.int64)
    dst = np.array([1, 0, 1, 2, 9], dtype=np.int64)
    cum = G.CumulativeCSR(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), 10, 10)
    cum.prefix(3, 3, 3)
    with pytest.raises(nv.SpmiError):
        cum.prefix(4, 3, 3)      # source 7 outside the first three nodes
    with pytest.raises(nv.SpmiError):
        cum.prefix(5, 8, 8)      # destination 9 outside the first eight nodes
    with pytest.raises(nv.SpmiError):
        cum.prefix(6, 10, 10)    # more edges than the table holds


def _dtdg_tables(n, e, t_steps, seed, n_feat=3):
    src, dst, hi = _growing_table(n, e, seed)
    e_time = np.minimum((np.arange(e) * t_steps) // e, t_steps - 1)
    n_time = np.minimum((np.arange(n) * t_steps) // n, t_steps - 1)
    # a node must exist no later than its first edge: move edge times forward as needed
    e_time = np.maximum(e_time, np.maximum(n_time[src], n_time[dst]))
    e_time = np.maximum.accumulate(e_time)
    rng = np.random.default_rng(seed + 100)
    edges = np.stack([e_time, src, dst], axis=1).astype(np.float64)
    nodes = np.concatenate([n_time[:, None].astype(np.float64), rng.standard_normal((n, n_feat))], axis=1)
    return edges, nodes, np.arange(t_steps)


def test_container_dispatch_is_the_same_either_way(dev):
    edges, nodes, ts = _dtdg_tables(4000, 60_000, 6, seed=11)
    a = VEInvariantDTDG(edges, nodes, ts, device=dev)
    b = VEInvariantDTDG(edges, nodes, ts, device=dev)
    b.cumulative = False
    for t in range(len(ts)):
        ga, gb = a.dispatcher(t)[0].observation, b.dispatcher(t)[0].observation
        assert ga._table is not None and gb._table is None
        for get in (lambda g: g.csr(want_eid=True), lambda g: g.csr_t(want_eid=True)):
            ca, cb = get(ga), get(gb)
            assert torch.equal(ca.indptr, cb.indptr) and torch.equal(ca.indices, cb.indices)
            assert torch.equal(ca.eid, cb.eid)
            assert (ca.n_heavy, ca.n_chunks) == (cb.n_heavy, cb.n_chunks)
        assert torch.equal(ga.in_degrees(), gb.in_degrees())
        fresh = a.dispatcher(t)[0].observation
        assert torch.equal(fresh.out_degrees(), gb.out_degrees())


def test_citation_feature_from_the_table_grouping(dev):
    edges, nodes, ts = _dtdg_tables(3000, 30_000, 5, seed=12, n_feat=2)
    a = CitationGraph(edges, nodes, ts, device=dev)
    b = CitationGraph(edges, nodes, ts, device=dev)
    b.cumulative = False
    for t in range(len(ts)):
        fa = a.dispatcher(t)[0].node_feature()
        fb = b.dispatcher(t)[0].node_feature()
        assert torch.equal(fa, fb)
        want = orc.citation_dispatch(edges, nodes, len(ts), t)[3]
        np.testing.assert_array_equal(fa.cpu().numpy(), want)


def test_bad_prefix_raises_before_any_gather_in_the_tracked_layer(dev):
    """A snapshot whose edges run ahead of its nodes: the prefix cut reports it through its
    status word and leaves part of `indices` unwritten.  The grad-enabled GraphConv builds a
    per-edge weight by gathering with those indices, so it must collect the verdict first and
    raise SpmiError instead of reading out of range."""
    from spiral_amd.core.types import GraphConv

    edges = np.array([[0, 1, 0], [0, 2, 1], [0, 7, 2], [1, 3, 1], [1, 8, 0]], dtype=np.float64)
    nodes = np.concatenate([np.array([0, 0, 0, 1, 1, 1, 1, 1, 1, 1])[:, None].astype(np.float64),
                            np.random.default_rng(0).standard_normal((10, 4))], axis=1)
    graph = VEInvariantDTDG(edges, nodes, np.arange(2), device=dev)
    snap, _ = graph.dispatcher(0)                       # 3 nodes, but an edge from node 7
    conv = GraphConv(4, 4, allow_zero_in_degree=True).to(dev)
    x = snap.node_feature().clone().requires_grad_(True)
    with pytest.raises(nv.SpmiError):
        conv(snap.observation, x)
    with torch.no_grad(), pytest.raises(nv.SpmiError):  # the inference path as well
        conv(graph.dispatcher(0)[0].observation, x.detach())
    good, _ = graph.dispatcher(1)
    out = conv(good.observation, good.node_feature())   # the table itself is fine
    assert out.shape == (10, 4) and bool(torch.isfinite(out).all())

"""GPU parity of the integer half of the path: degrees and COO -> CSR must be bit-exact
against the CPU oracle (stable grouping, eid = original position)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from spiral_amd import _native as nv
from spiral_amd import graph as G

pytestmark = pytest.mark.gpu


def _check_csr(src, dst, n, dev):
    csr = G.build_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n, want_eid=True)
    indptr, indices, eid = orc.coo_to_csr(dst, src, n)
    assert np.array_equal(csr.indptr.cpu().numpy(), indptr)
    assert np.array_equal(csr.indices.cpu().numpy(), indices)
    assert np.array_equal(csr.eid.cpu().numpy(), eid)
    lean = G.build_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n)
    assert lean.eid is None  # edge ids travel through the sort only on request
    assert np.array_equal(lean.indptr.cpu().numpy(), indptr)
    assert np.array_equal(lean.indices.cpu().numpy(), indices)
    return csr


@pytest.mark.parametrize("n,e,seed", [(1, 5, 0), (2, 1, 1), (7, 64, 2), (300, 4095, 3), (300, 4096, 4),
                                      (300, 4097, 5), (1000, 8000, 6), (70000, 200000, 7),
                                      (20_000_000, 300000, 8)])
def test_coo_to_csr_random(dev, n, e, seed):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e, dtype=np.int64)
    dst = rng.integers(0, n, e, dtype=np.int64)
    _check_csr(src, dst, n, dev)


def test_coo_to_csr_powerlaw_and_degrees(dev):
    n, e = 100_000, 1_000_000
    src, dst = orc.powerlaw_snapshot(n, e, t=0)
    csr = _check_csr(src, dst, n, dev)
    assert np.array_equal(csr.degrees().cpu().numpy(), orc.degree(dst, n))
    out_deg = G.degree(torch.from_numpy(src).to(dev), n)
    assert out_deg.dtype == torch.int64
    assert np.array_equal(out_deg.cpu().numpy(), orc.degree(src, n))


def test_coo_to_csr_edge_cases(dev):
    # no edges at all
    csr = G.build_csr(torch.zeros(0, dtype=torch.int64, device=dev),
                      torch.zeros(0, dtype=torch.int64, device=dev), 5, 5)
    assert csr.indptr.cpu().tolist() == [0] * 6
    # every edge into one node; long runs of empty rows on both sides of it
    n, e = 50_000, 20_000
    dst = np.full(e, 31_337, dtype=np.int64)
    src = np.arange(e, dtype=np.int64) % n
    _check_csr(src, dst, n, dev)
    # only the last / only the first node
    _check_csr(src, np.full(e, n - 1, dtype=np.int64), n, dev)
    _check_csr(src, np.zeros(e, dtype=np.int64), n, dev)
    # the reference fixture (tests/dtdg/test_types_torch.py:12): edges [t, src, dst]
    edges = np.array([[0, 0, 1], [0, 0, 2], [1, 3, 0], [2, 4, 1]])
    _check_csr(edges[:, 1].astype(np.int64), edges[:, 2].astype(np.int64), 5, dev)


def test_out_of_range_id_is_reported(dev):
    dst = torch.tensor([0, 1, 9], dtype=torch.int64, device=dev)
    src = torch.tensor([0, 1, 2], dtype=torch.int64, device=dev)
    with pytest.raises(nv.SpmiError):
        G.build_csr(dst, src, 5, 5)
    with pytest.raises(nv.SpmiError):
        G.build_csr(src, dst, 5, 5)  # bad id on the `other` side
    with pytest.raises(nv.SpmiError):
        G.degree(torch.tensor([-1, 2], dtype=torch.int64, device=dev), 5)


@pytest.mark.parametrize("n,e", [(1, 7), (4096, 50_000), (4097, 50_000), (300_000, 2_000_000), (1 << 20, 3_000_000),
                                 ((1 << 20) + 1, 3_000_000), (7_000_000, 12_000_000), (1 << 24, 3_000_000),
                                 ((1 << 24) + 1, 3_000_000)])
def test_degree_bucket_and_sort_paths(dev, n, e):
    """Up to 2^20 nodes degrees come from one grouping pass into buckets of 4096 nodes + LDS
    counters, up to 2^24 from two such passes (the second stable) and the bucket starts, above
    from the keys-only sort; all must equal a bincount, hubs included; the same for the
    normaliser counted from a 32-bit id column."""
    rng = np.random.default_rng(n + e)
    ids = rng.integers(0, n, e, dtype=np.int64)
    ids[: e // 3] = ids[0]                      # a hub holding a third of all entries
    ids[e // 3: e // 2] = n - 1                 # and the last node
    want = np.bincount(ids, minlength=n)
    got = G.degree(torch.from_numpy(ids).to(dev), n).cpu().numpy()
    assert np.array_equal(got, want)
    norm = G.norm_from_ids32(torch.from_numpy(ids.astype(np.int32)).to(dev), n, 0)
    ref = 1.0 / torch.from_numpy(want).to(dev).to(torch.float32).clamp(min=1).sqrt()
    assert torch.equal(norm, ref)


def test_deferred_build_reports_at_first_use(dev):
    """A CSR built with defer=True hands back control before the device has answered; the
    verdict on the ids arrives with ready(), and so with any aggregation on the graph."""
    good_src = torch.tensor([0, 1, 2], dtype=torch.int64, device=dev)
    good_dst = torch.tensor([1, 2, 4], dtype=torch.int64, device=dev)
    csr = G.build_csr(good_dst, good_src, 5, 5, defer=True)
    assert csr._pending is not None
    assert (csr.n_heavy, csr.n_chunks) == (0, 0) and csr._pending is None
    bad = G.build_csr(torch.tensor([0, 1, 9], dtype=torch.int64, device=dev), good_src, 5, 5, defer=True)
    with pytest.raises(nv.SpmiError):
        bad.ready()
    g = G.SnapshotGraph(good_src, torch.tensor([0, 1, 9], dtype=torch.int64, device=dev), 5)
    x = torch.ones(5, 4, device=dev)
    from spiral_amd import ops
    with pytest.raises(nv.SpmiError), torch.no_grad():
        ops.gcn_aggregate(g, x, "both")
    g2 = G.SnapshotGraph(torch.tensor([0, 7, 2], dtype=torch.int64, device=dev), good_dst, 5)
    with pytest.raises(nv.SpmiError), torch.no_grad():
        ops.gcn_aggregate(g2, x, "both")     # the bad id sits on the source side
    with pytest.raises(nv.SpmiError):
        g2.csr()                              # the waiting accessor reports it too


def test_norm_from_degree(dev):
    deg = torch.tensor([0, 1, 2, 4, 9, 1_000_000], dtype=torch.int64, device=dev)
    want = deg.to(torch.float32).clamp(min=1)
    assert torch.equal(G.norm_from_degree(deg, 0), 1.0 / want.sqrt())
    assert torch.equal(G.norm_from_degree(deg, 1), 1.0 / want)


def test_plan_lists_exactly_the_heavy_rows(dev):
    n, e = 20_000, 400_000
    src, dst = orc.powerlaw_snapshot(n, e, t=1, k_dst=6.0)
    csr = G.build_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n,
                      heavy_thresh=128, chunk=64)
    deg = orc.degree(dst, n)
    want = np.nonzero(deg > 128)[0]
    got = csr.heavy_rows[: csr.n_heavy].cpu().numpy()
    assert np.array_equal(np.sort(got), want)
    ptr = csr.heavy_chunk_ptr[: csr.n_heavy + 1].cpu().numpy()
    assert ptr[0] == 0 and ptr[-1] == csr.n_chunks
    assert np.array_equal(np.diff(ptr), (deg[got] + 63) // 64)


def test_norm_matches_oracle(dev):
    n, e = 5000, 40_000
    src, dst = orc.powerlaw_snapshot(n, e, t=2)
    g = G.SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
    ip_in, _, _ = orc.coo_to_csr(dst, src, n)
    ip_out, _, _ = orc.coo_to_csr(src, dst, n)
    for side, ip in (("in", ip_in), ("out", ip_out)):
        for mode, name in ((0, "rsqrt"), (1, "recip")):
            with torch.no_grad():
                got = g.norm(side, mode).cpu().numpy()
            np.testing.assert_allclose(got, orc.norm_from_indptr(ip, name), rtol=2e-7, atol=0)


@pytest.mark.parametrize("n,e", [(1000, 20_000), (1 << 20, 300_000), ((1 << 20) + 5, 200_000), (3_000_000, 400_000), (7, 0)])
@pytest.mark.parametrize("mode", [0, 1])
def test_normaliser_from_a_32_bit_id_column(dev, n, e, mode):
    """spmi_norm_from_ids32 (the source normaliser counted from the by-destination CSR's own
    int32 column) equals clamp(bincount, 1)^-1/2 resp. its reciprocal bit for bit with the
    kernels that go through int64 degrees, and the counts equal the oracle's degrees."""
    from spiral_amd import _native as nv

    rng = np.random.default_rng(n + e)
    ids = rng.integers(0, n, e).astype(np.int32)
    if e:
        ids[: e // 3] = ids[0]                                   # a hub
    t = torch.from_numpy(ids).to(dev)
    got = G.norm_from_ids32(t, n, mode)
    want_deg = orc.degree(ids.astype(np.int64), n)
    want = G.norm_from_degree(torch.from_numpy(want_deg).to(dev), mode)
    assert torch.equal(got, want)
    lib = nv.lib()
    deg = torch.full((n,), -1, dtype=torch.int64, device=dev)
    out = torch.empty(n, device=dev)
    nbytes = lib.spmi_degree_workspace_bytes(e, n)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    nv.check(lib.spmi_norm_from_ids32(nv.ptr(t), e, n, mode, nv.ptr(out), nv.ptr(deg), nv.ptr(ws), nbytes,
                                      nv.stream()), "spmi_norm_from_ids32")
    assert np.array_equal(deg.cpu().numpy(), want_deg) and torch.equal(out, want)


def _old_three_calls(key, other, n, n_other, want_eid, mode, dev):
    """spmi_coo_to_csr + spmi_norm_from_indptr + spmi_spmm_plan, the calls the one-call build
    must reproduce bit for bit."""
    lib = nv.lib()
    e = key.numel()
    indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    indices = torch.empty(e, dtype=torch.int32, device=dev)
    eid = torch.empty(e, dtype=torch.int32, device=dev) if want_eid else None
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.spmi_coo_to_csr_workspace_bytes(e, n)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    nv.check(lib.spmi_coo_to_csr(nv.ptr(key), nv.ptr(other), e, n, n_other, nv.ptr(indptr), nv.ptr(indices),
                                 nv.ptr(eid), nv.ptr(ws), ws_bytes, nv.ptr(status), nv.stream()), "coo_to_csr")
    norm = torch.empty(n, dtype=torch.float32, device=dev)
    nv.check(lib.spmi_norm_from_indptr(nv.ptr(indptr), n, mode, nv.ptr(norm), nv.stream()), "norm")
    cap = e // G.HEAVY_THRESH + 2
    rows = torch.empty(cap, dtype=torch.int32, device=dev)
    ptr = torch.empty(cap, dtype=torch.int32, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    nv.check(lib.spmi_spmm_plan(nv.ptr(indptr), n, G.HEAVY_THRESH, G.HEAVY_CHUNK, nv.ptr(rows), nv.ptr(ptr), cap,
                                nv.ptr(counts), nv.stream()), "plan")
    return indptr, indices, eid, norm, rows, ptr, counts.tolist(), int(status.item())


@pytest.mark.parametrize("n,e,k_dst,mode", [(4097, 30_000, 3.0, 0), (5000, 8192, 1.0, 1), (70_000, 400_001, 3.0, 0),
                                            (300_000, 2_000_000, 6.0, 0), (1 << 20, 3_000_000, 3.0, 1),
                                            (1_000_000, 10_000_000, 3.0, 0), (200_000, 1_000_000, 40.0, 0),
                                            # more than 256 buckets: two passes on the top bits, segment
                                            # layout by its own kernel
                                            ((1 << 20) + 1, 3_000_000, 3.0, 0), (3_000_000, 5_000_000, 6.0, 1),
                                            (10_000_000, 20_000_000, 3.0, 0), (1 << 24, 4_000_000, 3.0, 0),
                                            ((1 << 24) + 1, 4_000_000, 3.0, 0)])
def test_one_call_build_equals_the_three_calls(dev, n, e, k_dst, mode):
    """spmi_snapshot_csr on its bucket path (4096 < N <= 2^24: one or two radix passes on the top
    bits + the sort inside buckets of 4096 rows) against the radix sort on every bit, the normaliser kernel and the plan
    kernels: row pointers, column ids, edge ids and the normaliser bit for bit, the same set of
    hub rows with the same chunk counts (their order is free).  k_dst = 40 puts most edges on a
    handful of rows (buckets of many segments, empty buckets beside them)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(n + e)
    perm = torch.randperm(n, generator=gen, device=dev)
    dst = perm[(torch.rand(e, generator=gen, device=dev, dtype=torch.float64).pow_(k_dst) * n).long().clamp_(max=n - 1)]
    src = perm[(torch.rand(e, generator=gen, device=dev, dtype=torch.float64).pow_(1.5) * n).long().clamp_(max=n - 1)]
    for want_eid in (False, True):
        indptr, indices, eid, norm, rows, ptr, counts, status = _old_three_calls(dst, src, n, n, want_eid, mode, dev)
        assert status == 0
        csr = G.build_csr(dst, src, n, n, want_eid=want_eid, norm_mode=mode, other_norm=True)
        assert torch.equal(csr.indptr, indptr) and torch.equal(csr.indices, indices)
        # the other side's normaliser, made by the same pass where the library offers it
        assert (csr.other_norm(mode) is not None) == bool(nv.lib().spmi_snapshot_csr_norm_other_supported(e, n, n))
        if csr.other_norm(mode) is not None:
            d = torch.bincount(src, minlength=n).to(torch.float32).clamp_(min=1)
            assert torch.equal(csr.other_norm(mode), 1.0 / d.sqrt() if mode == 0 else 1.0 / d)
        assert (csr.eid is None) == (not want_eid) and (eid is None or torch.equal(csr.eid, eid))
        assert torch.equal(csr.norm(mode), norm)
        assert [csr.n_heavy, csr.n_chunks] == counts
        nh = counts[0]
        got = torch.sort(csr.heavy_rows[:nh])
        want = torch.sort(rows[:nh])
        assert torch.equal(got.values, want.values)
        chunks = lambda p: (p[1: nh + 1] - p[:nh]) if nh else p[:0]
        assert torch.equal(chunks(csr.heavy_chunk_ptr)[got.indices], chunks(ptr)[want.indices])
        assert int(csr.heavy_chunk_ptr[nh]) == counts[1]
    if e <= 3_000_000:   # and the oracle itself
        ip, ix, ei = orc.coo_to_csr(dst.cpu().numpy(), src.cpu().numpy(), n)
        assert np.array_equal(csr.indptr.cpu().numpy(), ip) and np.array_equal(csr.indices.cpu().numpy(), ix)
        assert np.array_equal(csr.eid.cpu().numpy(), ei)


def test_one_call_build_reports_bad_ids_and_runs_twice(dev):
    n = 10_000
    dst = torch.randint(0, n, (50_000,), device=dev)
    src = torch.randint(0, n, (50_000,), device=dev)
    bad = dst.clone()
    bad[123] = n
    with pytest.raises(nv.SpmiError):
        G.build_csr(bad, src, n, n)
    with pytest.raises(nv.SpmiError):
        G.build_csr(src, bad, n, n)          # bad id on the `other` side
    a = G.build_csr(dst, src, n, n, want_eid=True, norm_mode=0)
    b = G.build_csr(dst, src, n, n, want_eid=True, norm_mode=0)
    assert torch.equal(a.indices, b.indices) and torch.equal(a.eid, b.eid) and torch.equal(a.norm(0), b.norm(0))


def test_builds_on_a_side_stream(dev):
    """Snapshot builds queued on a stream other than the one an earlier build ran on: the
    zeroed status / plan words of a build belong to the stream that is current (a block of such
    words zeroed on one stream is never handed to another), so the range check and the hub-row
    plan are what they are on the default stream."""
    n, e = 50_000, 400_000
    src, dst = orc.powerlaw_snapshot(n, e, t=3, k_dst=6.0)
    src_t, dst_t = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
    ref = G.build_csr(dst_t, src_t, n, n, want_eid=True)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    bad = dst_t.clone()
    bad[7] = n + 5
    with torch.cuda.stream(side):
        for _ in range(3):
            got = G.build_csr(dst_t, src_t, n, n, want_eid=True)
            assert (got.n_heavy, got.n_chunks) == (ref.n_heavy, ref.n_chunks) and ref.n_heavy > 0
            assert torch.equal(got.indptr, ref.indptr) and torch.equal(got.indices, ref.indices)
        with pytest.raises(nv.SpmiError):
            G.build_csr(bad, src_t, n, n)
    side.synchronize()
    again = G.build_csr(dst_t, src_t, n, n)
    assert (again.n_heavy, again.n_chunks) == (ref.n_heavy, ref.n_chunks)

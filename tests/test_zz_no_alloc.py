"""spmi_gcn_spmm_static (the aggregation entry point that allocates nothing: hub-row partial
sums go to an array built into the library) against spmi_gcn_spmm_ex with a caller
workspace, both called through the C-ABI.  Same kernels, so the bar is bit equality.
Two calls in a row, the second on a smaller graph, so that what the first call left in the
built-in array must not leak into the second result."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from spiral_amd import _native as nv
from spiral_amd import graph as G

pytestmark = pytest.mark.gpu

STATIC_FLOATS = 4 * 1024 * 1024  # SPMI_STATIC_PARTIAL_FLOATS of include/spmi.h


def _call(name, csr, x, ds, bias, y, partial=None):
    args = [nv.ptr(csr.indptr), nv.ptr(csr.indices), csr.n_rows, csr.n_cols, csr.nnz, nv.ptr(x), 0,
            x.stride(0), x.shape[1], None, None, nv.ptr(ds), nv.ptr(bias), 2, 0.01, nv.ptr(y), 0,
            y.stride(0), nv.ptr(csr.heavy_rows), nv.ptr(csr.heavy_chunk_ptr), csr.n_heavy,
            csr.n_chunks, csr.heavy_thresh, csr.chunk]
    if name == "spmi_gcn_spmm_ex":
        args.append(nv.ptr(partial))
    args.append(nv.stream())
    return getattr(nv.lib(), name)(*args)


def _case(dev, n, e, f, seed):
    src, dst = orc.powerlaw_snapshot(n, e, t=seed, k_dst=6.0)
    csr = G.build_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n)
    assert csr.n_heavy > 0 and csr.n_chunks * f <= STATIC_FLOATS
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n, f, generator=g).to(dev)
    ds = torch.rand(n, generator=g).to(dev)
    bias = torch.randn(f, generator=g).to(dev)
    return src, dst, csr, x, ds, bias


def test_static_scratch_matches_workspace_entry_point(dev):
    results = []
    for n, e, f, seed in ((40_000, 1_200_000, 128, 21), (9_000, 150_000, 64, 22)):
        src, dst, csr, x, ds, bias = _case(dev, n, e, f, seed)
        want = torch.full((n, f), float("nan"), device=dev)
        got = torch.full((n, f), float("nan"), device=dev)
        partial = torch.empty(csr.n_chunks * f, dtype=torch.float32, device=dev)
        nv.check(_call("spmi_gcn_spmm_ex", csr, x, ds, bias, want, partial), "spmi_gcn_spmm_ex")
        nv.check(_call("spmi_gcn_spmm_static", csr, x, ds, bias, got), "spmi_gcn_spmm_static")
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"static-scratch result differs at n={n}"
        results.append((csr.n_heavy, csr.n_chunks))
        if n <= 10_000:  # the small case also against the oracle, with the usual 1e-5 bar
            indptr, indices, _ = orc.coo_to_csr(dst, src, n)
            ref = orc.aggregate(indptr, indices, x.cpu().numpy(), None, None, ds.cpu().numpy(),
                                bias.cpu().numpy(), "leaky_relu", 0.01)
            scale = orc.aggregate(indptr, indices, np.abs(x.cpu().numpy()), None, None,
                                  ds.cpu().numpy(), np.abs(bias.cpu().numpy())) + 1e-30
            assert (np.abs(got.cpu().numpy() - ref) / scale).max() <= 1e-5
    assert results[1][1] < results[0][1], "second call must use less of the scratch than the first"
    # a warmed-up call takes nothing from the device: free memory is the same before and after
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info(dev)
    nv.check(_call("spmi_gcn_spmm_static", csr, x, ds, bias, got), "spmi_gcn_spmm_static")
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_after == free_before, (free_before, free_after)


def test_static_scratch_rejects_plans_beyond_its_size(dev):
    n, f = 64, 128
    indptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    y = torch.empty(n, f, device=dev)
    n_chunks = STATIC_FLOATS // f + 1
    rc = nv.lib().spmi_gcn_spmm_static(
        nv.ptr(indptr), nv.ptr(indptr), n, n, 0, nv.ptr(y), 0, f, f, None, None, None, None, 0, 0.0,
        nv.ptr(y), 0, f, nv.ptr(indptr), nv.ptr(indptr), 1, n_chunks, 128, 256, nv.stream())
    assert rc == -3 and b"scratch" in nv.lib().spmi_last_error()

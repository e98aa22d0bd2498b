"""The weight-gradient GEMM C = A^T B (split-K on the matrix cores; products from exact bf16
pieces or on the fp32 instruction) against float64 numpy: fp32-accurate products and fp32
sums, so 1e-5 relative to the sum of absolute terms."""
import numpy as np
import pytest
import torch

from spiral_amd import _native as nv
from spiral_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,m,n", [
    (16384, 128, 128), (20001, 128, 384), (50_000, 10, 128), (33_333, 20, 10), (17_000, 5, 1),
    (16_400, 33, 65), (100_000, 256, 512), (16385, 1, 1),
])
@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
def test_gemm_tn_matches_float64(dev, k, m, n, mode, monkeypatch):
    monkeypatch.setattr(ops, "GEMM_TN_MODE", mode)
    rng = np.random.default_rng(k + m + n)
    a = rng.standard_normal((k, m)).astype(np.float32)
    b = rng.standard_normal((k, n)).astype(np.float32)
    got = ops.gemm_tn(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    want = a.astype(np.float64).T @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)
    assert got.shape == (m, n)
    assert (np.abs(got - want) / scale).max() < 1e-5
    again = ops.gemm_tn(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    assert np.array_equal(got, again)  # fixed summation order


def test_gemm_tn_strided_operands_and_accumulate(dev):
    k, m, n = 40_000, 48, 96
    big_a = torch.randn(k, 3 * m, device=dev)
    big_b = torch.randn(k, 2 * n, device=dev)
    a, b = big_a[:, m:2 * m], big_b[:, n:]           # column slices: leading dimension > width
    want = a.double().t() @ b.double()
    got = ops.gemm_tn(a, b)
    assert (got.double() - want).abs().max() / want.abs().max() < 1e-5
    lib = nv.lib()
    c = torch.ones(m, n, device=dev)
    nbytes = lib.spmi_gemm_tn_workspace_bytes(m, n, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    nv.check(lib.spmi_gemm_tn(nv.ptr(a), a.stride(0), nv.ptr(b), b.stride(0), k, m, n, nv.ptr(c), n, 1,
                              nv.ptr(ws), nbytes, nv.stream()), "spmi_gemm_tn")
    assert (c.double() - 1 - want).abs().max() / want.abs().max() < 1e-5
    with pytest.raises(nv.SpmiError):
        nv.check(lib.spmi_gemm_tn(nv.ptr(a), a.stride(0), nv.ptr(b), b.stride(0), k, m, n, nv.ptr(c), n, 0,
                                  nv.ptr(ws), 16, nv.stream()), "spmi_gemm_tn")


def test_dense_layers_use_it_and_keep_their_gradients(dev):
    """TallLinear / rows_times_weight: same forward as torch, gradients within 1e-5."""
    from spiral_amd.nn.linear import TallLinear

    k = 30_000
    torch.manual_seed(0)
    lin = TallLinear(24, 40).to(dev)
    ref = torch.nn.Linear(24, 40).to(dev)
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(k, 24, device=dev)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    wgt = torch.randn(k, 40, device=dev)
    (lin(xa) * wgt).sum().backward()
    (ref(xb) * wgt).sum().backward()
    rel = lambda p, q: ((p - q).abs().max() / q.abs().max()).item()
    assert rel(xa.grad, xb.grad) < 1e-5
    assert rel(lin.weight.grad, ref.weight.grad) < 1e-5
    assert rel(lin.bias.grad, ref.bias.grad) < 1e-5
    w = torch.randn(24, 40, device=dev)
    wa, wb = w.clone().requires_grad_(), w.clone().requires_grad_()
    (ops.rows_times_weight(x, wa) * wgt).sum().backward()
    ((x @ wb) * wgt).sum().backward()
    assert rel(wa.grad, wb.grad) < 1e-5


@pytest.mark.parametrize("k,m,n", [(20001, 128, 384), (50_000, 10, 128), (16_400, 33, 65), (100_000, 384, 128),
                                   (17_000, 5, 1)])
def test_operand_column_sums_ride_along(dev, k, m, n):
    """spmi_gemm_tn_sums: the column sums of A and of B from the same pass (bias gradients next
    to weight gradients), float64 reference, 1e-5 relative to the sum of absolute terms, fixed
    order; the product is unchanged bit for bit."""
    rng = np.random.default_rng(k + m)
    a = (rng.standard_normal((k, m)) + 0.3).astype(np.float32)
    b = (rng.standard_normal((k, n)) - 0.2).astype(np.float32)
    at, bt = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    got, sa, sb = ops.gemm_tn(at, bt, sum_a=True, sum_b=True)
    assert torch.equal(got, ops.gemm_tn(at, bt))
    for s, ref in ((sa, a), (sb, b)):
        want = ref.astype(np.float64).sum(0)
        scale = np.abs(ref).astype(np.float64).sum(0)
        assert s.shape == want.shape
        assert (np.abs(s.cpu().numpy() - want) / scale).max() < 1e-5
    _, sa2, _ = ops.gemm_tn(at, bt, sum_a=True)
    assert torch.equal(sa, sa2)


def test_bf16_pieces_keep_fp32_range_and_accuracy(dev):
    """Operands of very different magnitudes (gradients of a mean loss over a million rows are
    1e-6 and smaller): the bf16-piece products carry no range limit and are as close to float64
    as the fp32 instruction's."""
    k, m, n = 60_000, 96, 64
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((k, m)) * np.exp(rng.uniform(-30, 5, (k, 1)))).astype(np.float32)
    b = (rng.standard_normal((k, n)) * 1e-7).astype(np.float32)
    at, bt = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    want = a.astype(np.float64).T @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)
    errs = {}
    for mode in ("bf16x3", "fp32"):
        old, ops.GEMM_TN_MODE = ops.GEMM_TN_MODE, mode
        try:
            errs[mode] = (np.abs(ops.gemm_tn(at, bt).cpu().numpy() - want) / scale).max()
        finally:
            ops.GEMM_TN_MODE = old
    assert errs["bf16x3"] < 1e-5 and errs["bf16x3"] < 4 * errs["fp32"] + 1e-7, errs


@pytest.mark.parametrize("k", [1, 5, 16, 17, 31, 48, 300, 4101])
def test_whole_block_kernel_at_short_and_ragged_k(dev, k):
    """Straight through the C-ABI (the Python wrapper sends short matrices to the library):
    128 x 128 and 256 x 128 outputs, i.e. the kernel that owns whole blocks, for reductions
    shorter than one K-step, ending inside a step, and with an odd number of steps; leading
    dimensions wider than the operands; column sums included."""
    lib = nv.lib()
    rng = np.random.default_rng(k)
    for m, n in ((128, 128), (256, 128)):
        big_a = rng.standard_normal((k, m + 4)).astype(np.float32)
        big_b = rng.standard_normal((k, n + 8)).astype(np.float32)
        ta, tb = torch.from_numpy(big_a).to(dev), torch.from_numpy(big_b).to(dev)
        a, b = ta[:, 4:], tb[:, :n]                       # 16-byte aligned column ranges
        c = torch.empty(m, n, device=dev)
        sa, sb = torch.empty(m, device=dev), torch.empty(n, device=dev)
        nbytes = lib.spmi_gemm_tn_workspace_bytes(m, n, k)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        nv.check(lib.spmi_gemm_tn_sums(nv.ptr(a), a.stride(0), nv.ptr(b), b.stride(0), k, m, n, nv.ptr(c), n, 0,
                                       nv.ptr(sa), nv.ptr(sb), 0, nv.ptr(ws), nbytes, nv.stream()),
                 "spmi_gemm_tn_sums")
        want = big_a[:, 4:].astype(np.float64).T @ big_b[:, :n].astype(np.float64)
        scale = np.abs(big_a[:, 4:]).astype(np.float64).T @ np.abs(big_b[:, :n]).astype(np.float64)
        assert (np.abs(c.cpu().numpy() - want) / scale).max() < 1e-5, (k, m, n)
        for got, ref in ((sa, big_a[:, 4:]), (sb, big_b[:, :n])):
            err = np.abs(got.cpu().numpy() - ref.astype(np.float64).sum(0))
            assert (err / np.abs(ref).astype(np.float64).sum(0)).max() < 1e-5, (k, m, n)


@pytest.mark.parametrize("k", [1, 17, 300, 20_001, 100_000])
def test_two_products_in_one_launch(dev, k):
    """spmi_gemm_tn_pair_sums straight through the C-ABI: two products over overlapping column
    ranges of one [k, 512] matrix (the packed gate gradients of a GRU layer: columns 0..383 with
    x, columns 128..511 with h), each against float64 at 1e-5 of the sum of absolute terms, with
    the column sums of both ranges; repeated launches agree bit for bit."""
    lib = nv.lib()
    rng = np.random.default_rng(k)
    g4 = (rng.standard_normal((k, 512)) * 1e-3).astype(np.float32)
    x = rng.standard_normal((k, 128)).astype(np.float32)
    h = (rng.standard_normal((k, 128)) * 0.5).astype(np.float32)
    tg, tx, th = (torch.from_numpy(t).to(dev) for t in (g4, x, h))
    a0, a1 = tg[:, :384], tg[:, 128:]
    m, n = 384, 128
    assert lib.spmi_gemm_tn_pair_supported(m, n) and not lib.spmi_gemm_tn_pair_supported(100, 128)
    nbytes = lib.spmi_gemm_tn_pair_workspace_bytes(m, n, k)

    def run(f16_planes=1):
        c0, c1 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
        s0, s1 = torch.empty(m, device=dev), torch.empty(m, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        nv.check(lib.spmi_gemm_tn_pair_sums(
            nv.ptr(a0), a0.stride(0), nv.ptr(tx), 128, nv.ptr(a1), a1.stride(0), nv.ptr(th), 128, k, m, n,
            nv.ptr(c0), n, nv.ptr(c1), n, nv.ptr(s0), nv.ptr(s1), f16_planes, nv.ptr(ws), nbytes, nv.stream()),
            "spmi_gemm_tn_pair_sums")
        return c0, c1, s0, s1

    for planes in (0, 1):  # bf16 pieces, fp16 planes under running scales
        got = run(planes)
        for a, b, c, s in ((g4[:, :384], x, got[0], got[2]), (g4[:, 128:], h, got[1], got[3])):
            want = a.astype(np.float64).T @ b.astype(np.float64)
            scale = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)
            assert (np.abs(c.cpu().numpy() - want) / scale).max() < 1e-5, planes
            assert (np.abs(s.cpu().numpy() - a.astype(np.float64).sum(0)) / np.abs(a).astype(np.float64).sum(0)).max() < 1e-5
    again = run()
    assert all(torch.equal(p, q) for p, q in zip(got, again))
    if k >= ops.GEMM_TN_MIN_ROWS:
        assert ops.gemm_tn_pair_ok(a0, tx, a1, th)
        c0, c1 = ops.gemm_tn_pair(a0, tx, a1, th)
        assert torch.equal(c0, got[0]) and torch.equal(c1, got[1])


@pytest.mark.parametrize("case", ["wild_rows", "tiny_then_large", "zero_tiles", "gradient_sized", "large"])
def test_whole_block_fp16_planes_follow_the_operands_magnitudes(dev, case):
    """The fp16-plane arithmetic of the whole-block kernel (fp32_instruction = 2) scales its
    operands K-step by K-step: rows whose magnitudes differ by 2^50 inside one chunk, a chunk that
    starts 2^40 below where it ends (a second pass over the chunk takes the steps that did not fit),
    tiles of zeros, gradients of 1e-7 against activations of 1e3, values of 1e8; each against
    float64 at 1e-5 of the sum of absolute terms, with the bf16-piece arithmetic beside it."""
    lib = nv.lib()
    k, m, n = 40_000, 128, 256
    rng = np.random.default_rng(11)
    a = rng.standard_normal((k, m))
    b = rng.standard_normal((k, n))
    if case == "wild_rows":
        a *= np.exp(rng.uniform(-30, 5, (k, 1)))
        b *= np.exp(rng.uniform(-5, 5, (k, 1)))
    elif case == "tiny_then_large":
        ramp = np.where(np.arange(k) % 8000 < 48, 2.0 ** -40, 1.0)[:, None]   # every chunk region starts tiny
        a *= ramp
    elif case == "zero_tiles":
        a[: k // 2 : 1][(np.arange(k // 2) // 16) % 3 == 0] = 0.0
        b[(np.arange(k) // 16) % 5 == 1] = 0.0
    elif case == "gradient_sized":
        a *= 1e-7
        b *= 1e3
    else:
        a *= 1e8
        b *= 1e8
    a, b = a.astype(np.float32), b.astype(np.float32)
    at, bt = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    want = a.astype(np.float64).T @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)
    nbytes = lib.spmi_gemm_tn_workspace_bytes(m, n, k)
    errs = {}
    for mode in (0, 2):
        c = torch.empty(m, n, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        nv.check(lib.spmi_gemm_tn_sums(nv.ptr(at), m, nv.ptr(bt), n, k, m, n, nv.ptr(c), n, 0, None, None, mode,
                                       nv.ptr(ws), nbytes, nv.stream()), "spmi_gemm_tn_sums")
        got = c.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), (case, mode)
        errs[mode] = float((np.abs(got - want) / np.maximum(scale, 1e-300)).max())
    print(case, "bf16 pieces", errs[0], "fp16 planes", errs[2])
    assert errs[0] < 1e-5 and errs[2] < 1e-5, (case, errs)

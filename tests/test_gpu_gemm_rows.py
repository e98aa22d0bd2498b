"""The memory-speed projection kernel (spmi_gemm_rows) against float64 on the CPU: every
supported (K, N), ragged row counts, strided input, bias and activations, gradients of the
layers that use it, and the loud failure beyond fp16's range."""
import numpy as np
import pytest
import torch

from spiral_amd import _native as nv
from spiral_amd import ops

pytestmark = pytest.mark.gpu


def _ref(x, w_nk, bias, act, slope=0.01):
    y = x.double().cpu() @ w_nk.double().cpu().t()
    if bias is not None:
        y = y + bias.double().cpu()
    if act == "relu":
        y = y.clamp(min=0)
    elif act == "leaky_relu":
        y = torch.where(y > 0, y, y * slope)
    return y


@pytest.mark.parametrize("k", [32, 64, 128])
@pytest.mark.parametrize("n", [32, 64, 96, 128])
def test_all_shapes(dev, k, n):
    gen = torch.Generator().manual_seed(k * 1000 + n)
    for m in (1, 31, 32, 33, 4096 + 77):
        x = (torch.randn(m, k, generator=gen) * 2).to(dev)
        w = (torch.randn(n, k, generator=gen) * 0.2).to(dev)
        b = torch.randn(n, generator=gen).to(dev)
        got = ops.gemm_rows(x, w, b)
        want = _ref(x, w, b, None)
        scale = float(want.abs().max())
        assert float((got.double().cpu() - want).abs().max()) < 2e-6 * scale, (m, k, n)


@pytest.mark.parametrize("k,n", [(256, 512), (256, 64), (256, 192), (128, 256), (64, 512), (128, 384),
                                 (384, 128), (384, 64), (384, 256), (512, 512), (512, 64)])
def test_wide_outputs_in_column_slices(dev, k, n):
    """N > 128 and K = 256, 384 (the 256 -> 512 projection of the
    wide-feature workload; the gate gradients [rows, 3 x 128] times W_ih of a GRU layer's
    backward): column slices computed by different workgroups, with bias, activation and row
    scale per slice; rows wider than 128 columns pass through the registers in chunks."""
    assert nv.lib().spmi_gemm_rows_supported(k, n)
    gen = torch.Generator().manual_seed(k + n)
    for m in (1, 33, 5000 + 13):
        x = (torch.randn(m, k, generator=gen) * 2).to(dev)
        w = (torch.randn(n, k, generator=gen) * 0.1).to(dev)
        b = torch.randn(n, generator=gen).to(dev)
        rs = (torch.rand(m, generator=gen) + 0.5).to(dev)
        got = ops.gemm_rows(x, w, b, "leaky_relu", 0.01, rs)
        want = _ref(x, w, b, "leaky_relu") * rs.double().cpu()[:, None]
        scale = float(want.abs().max())
        assert got.shape == (m, n)
        assert float((got.double().cpu() - want).abs().max()) < 2e-6 * scale, (m, k, n)
    assert not nv.lib().spmi_gemm_rows_supported(256, 96)
    assert not nv.lib().spmi_gemm_rows_supported(128, 160)
    assert not nv.lib().spmi_gemm_rows_supported(512, 96)
    assert not nv.lib().spmi_gemm_rows_supported(640, 128)


def test_closer_to_float64_than_the_library(dev):
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(200_000, 128, generator=gen) * 2).to(dev)
    w = (torch.randn(128, 128, generator=gen) * 0.2).to(dev)
    want = _ref(x, w, None, None)
    err_kernel = float((ops.gemm_rows(x, w).double().cpu() - want).abs().max())
    err_lib = float(((x @ w.t()).double().cpu() - want).abs().max())
    assert err_kernel <= err_lib * 1.5, (err_kernel, err_lib)


@pytest.mark.parametrize("act", [None, "relu", "leaky_relu"])
def test_strided_input_bias_and_activation(dev, act):
    gen = torch.Generator().manual_seed(11)
    wide = torch.randn(5000, 200, generator=gen).to(dev)
    x = wide[:, 8:136]                                   # row stride 200, 16-byte aligned start
    w = (torch.randn(64, 128, generator=gen) * 0.1).to(dev)
    b = torch.randn(64, generator=gen).to(dev)
    assert ops._gemm_rows_ok(x, 128, 64)
    got = ops.gemm_rows(x, w, b, act, 0.2)
    want = _ref(x, w, b, act, 0.2)
    assert float((got.double().cpu() - want).abs().max()) < 1e-5
    assert not ops._gemm_rows_ok(wide[:, 3:131], 128, 64)  # misaligned start: library path


def test_layers_route_through_it_and_gradients_match(dev):
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(6000, 128, generator=gen).to(dev).requires_grad_()
    w = (torch.randn(128, 64, generator=gen) * 0.1).to(dev).requires_grad_()
    b = torch.randn(64, generator=gen).to(dev).requires_grad_()
    calls = []
    real = ops.gemm_rows
    ops.gemm_rows = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        y = ops.rows_times_weight(x, w, b)
        y.square().sum().backward()
        lin_w = (torch.randn(32, 128, generator=gen) * 0.1).to(dev).requires_grad_()
        z = ops.linear_rows(x.detach().requires_grad_(), lin_w, None)
        z.sum().backward()
    finally:
        ops.gemm_rows = real
    assert len(calls) == 4                                # two forwards, two input gradients
    xd, wd, bd = (t.detach().double().cpu().requires_grad_() for t in (x, w, b))
    (xd @ wd + bd).square().sum().backward()
    for got, want in ((x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        scale = float(want.abs().max())
        assert float((got.double().cpu() - want).abs().max()) < 1e-5 * scale


def test_rows_of_any_magnitude_and_bad_shapes(dev):
    """Rows are scaled by a power of two before they are split, so values far outside fp16's
    range (degree counts, unnormalised features) and tiny rows come out at fp32 accuracy;
    non-finite input stays non-finite in its own row only."""
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(4096, 32, generator=gen)
    x[100] *= 7.0e4
    x[101] *= 1.0e30
    x[102] *= 1.0e-30
    x[103, 5] = 3.0e9                      # one huge entry among ordinary ones
    x[104] = 0.0
    x = x.to(dev)
    w = (torch.randn(32, 32, generator=gen) * 0.1).to(dev)
    out = ops.gemm_rows(x, w).double().cpu()
    want = _ref(x, w, None, None)
    scale = want.abs().amax(dim=1, keepdim=True).clamp(min=1e-300)
    assert torch.isfinite(out).all()
    assert float(((out - want).abs() / scale).max()) < 2e-6
    assert float(out[104].abs().max()) == 0.0
    x[200, 3] = float("inf")
    x[201, 0] = float("nan")
    out = ops.gemm_rows(x, w)
    assert not torch.isfinite(out[200]).any() and not torch.isfinite(out[201]).any()
    keep = torch.ones(4096, dtype=torch.bool)
    keep[200] = keep[201] = False
    assert torch.isfinite(out[keep.to(dev)]).all()
    # rows wider than 128 columns are scaled chunk by chunk: chunks of very different size in
    # one row, a huge row, a tiny row, a non-finite value in the second chunk only
    xw = torch.randn(4100, 256, generator=gen)
    xw[7, :128] *= 1.0e-20
    xw[7, 128:] *= 1.0e20
    xw[8] *= 3.0e30
    xw[9] *= 1.0e-30
    xw[10, 128:] = 0.0
    xw = xw.to(dev)
    ww = (torch.randn(64, 256, generator=gen) * 0.1).to(dev)
    outw = ops.gemm_rows(xw, ww).double().cpu()
    wantw = _ref(xw, ww, None, None)
    scalew = wantw.abs().amax(dim=1, keepdim=True).clamp(min=1e-300)
    assert torch.isfinite(outw).all()
    assert float(((outw - wantw).abs() / scalew).max()) < 2e-6
    xw[11, 200] = float("inf")
    outw = ops.gemm_rows(xw, ww)
    assert not torch.isfinite(outw[11]).any()
    keepw = torch.ones(4100, dtype=torch.bool)
    keepw[11] = False
    assert torch.isfinite(outw[keepw.to(dev)]).all()
    assert not nv.lib().spmi_gemm_rows_supported(48, 32)
    assert not nv.lib().spmi_gemm_rows_supported(128, 200)
    with pytest.raises(nv.SpmiError):
        nv.check(nv.lib().spmi_gemm_rows(nv.ptr(x), 32, 4096, 48, nv.ptr(w), 32, None, None, 0, 0.0,
                                         nv.ptr(out), 32, nv.stream()), "spmi_gemm_rows")


def test_row_scale(dev):
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(5003, 64, generator=gen).to(dev)
    w = (torch.randn(96, 64, generator=gen) * 0.1).to(dev)
    b = torch.randn(96, generator=gen).to(dev)
    rs = (torch.rand(5003, generator=gen) + 0.1).to(dev)
    got = ops.gemm_rows(x, w, b, "relu", 0.0, rs)
    want = _ref(x, w, b, "relu") * rs.double().cpu()[:, None]
    assert float((got.double().cpu() - want).abs().max()) < 1e-5
    with pytest.raises(nv.SpmiError):
        ops.gemm_rows(x, w, b, None, 0.0, rs[:100])


@pytest.mark.parametrize("norm", ["both", "left", "right", "none"])
def test_graphconv_inference_path_equals_tracked_path(dev, norm):
    """Without autograd GraphConv scales the projected rows in the GEMM and aggregates with no
    per-edge weight; with autograd it streams a per-edge weight.  Same numbers to 1e-5."""
    from oracle import oracle as orc
    from spiral_amd import graph as G
    from spiral_amd.core.types import GraphConv

    n, e = 6000, 50_000
    src, dst = orc.powerlaw_snapshot(n, e, t=3)
    g = G.SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
    torch.manual_seed(5)
    conv = GraphConv(64, 64, norm=norm, activation=torch.nn.functional.leaky_relu,
                     allow_zero_in_degree=True).to(dev)
    x = torch.randn(n, 64, device=dev)
    calls = []
    real = ops.graphconv_inference
    import spiral_amd.nn.graphconv as gc_mod
    gc_mod.graphconv_inference = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            fast = conv(g, x)
        tracked = conv(g, x)
    finally:
        gc_mod.graphconv_inference = real
    assert len(calls) == 1 and tracked.requires_grad and not fast.requires_grad
    scale = float(tracked.detach().abs().max())
    assert float((fast - tracked.detach()).abs().max()) < 1e-5 * max(scale, 1.0)


def test_weight_planes_cache_follows_the_weight(dev):
    """Planes are reused while the weight object is unchanged and rebuilt after an in-place
    update, a re-assignment of .data, or for another tensor at the same address."""
    w = torch.nn.Parameter(torch.randn(64, 64, device=dev) * 0.1)
    x = torch.randn(4096, 64, device=dev)
    p1 = ops.weight_planes(w, "f16x3", True)
    assert ops.weight_planes(w, "f16x3", True) is p1
    assert ops.weight_planes(w, "f16x3", False) is not p1            # other orientation: own entry
    with torch.no_grad():
        y1 = ops.rows_times_weight(x, w)
        w.mul_(2.0)                                                   # version bump
        y2 = ops.rows_times_weight(x, w)
        assert torch.allclose(y2, 2.0 * y1, rtol=1e-5, atol=1e-6)
        w.data = torch.randn(64, 64, device=dev) * 0.1                # new storage, same object
        y3 = ops.rows_times_weight(x, w)
    want = x.double().cpu() @ w.detach().double().cpu()
    assert float((y3.double().cpu() - want).abs().max()) < 1e-5
    n_before = len(ops._plane_cache)
    del w
    import gc
    gc.collect()
    assert len(ops._plane_cache) < n_before                           # entries die with the tensor


def test_edits_through_dot_data_need_an_explicit_invalidation(dev):
    """`.data` edits bump no version counter and move no storage, so the cached planes cannot
    notice them; `ops.invalidate_weight_caches()` (also called by train()/eval() and after
    load_state_dict of the package's modules) makes the next call re-split the weight."""
    from spiral_amd.nn.linear import TallLinear

    lin = TallLinear(64, 64, bias=False).to(dev)
    x = torch.randn(4096, 64, device=dev)
    with torch.no_grad():
        y1 = lin(x)
        lin.weight.data.mul_(2.0)
        ops.invalidate_weight_caches()
        y2 = lin(x)
        assert torch.allclose(y2, 2.0 * y1, rtol=1e-5, atol=1e-6)
        lin.weight.data.mul_(0.5)
        lin.eval()                                   # the module hook drops the planes as well
        y3 = lin(x)
        assert torch.allclose(y3, y1, rtol=1e-5, atol=1e-6)
        state = {k: v * 3.0 for k, v in lin.state_dict().items()}
        outer = torch.nn.Sequential(lin)
        outer.load_state_dict({"0." + k: v for k, v in state.items()})
        y4 = lin(x)
        assert torch.allclose(y4, 3.0 * y1, rtol=1e-5, atol=1e-5)


def test_weight_beyond_the_fp16_plane_domain_is_reported(dev):
    """|W| > 1023 makes spmi_gemm_rows return inf/NaN; the cached planes remember the weight's
    magnitude and a later use raises instead of going on with such results."""
    w = torch.nn.Parameter(torch.randn(64, 64, device=dev))
    with torch.no_grad():
        w[3, 5] = 5000.0
    x = torch.randn(4096, 64, device=dev)
    with torch.no_grad():
        ops.rows_times_weight(x, w)
        torch.cuda.synchronize()
        with pytest.raises(nv.SpmiError, match="domain"):
            ops.rows_times_weight(x, w)


@pytest.mark.parametrize("k,n", [(128, 128), (64, 96), (32, 32)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_rows_stored_as_bfloat16(dev, k, n, out_dtype):
    """spmi_gemm_rows_bf16: bfloat16 rows (exact operands) against the float32 weight kept as
    three bf16 pieces.  Reference: float64 on the stored row values; a float32 result is within
    1e-5 of the scale, a bfloat16 one within bfloat16's own rounding (2^-8) of it, where a
    bfloat16 GEMM would have rounded the weight as well."""
    assert nv.lib().spmi_gemm_rows_bf16_supported(k, n) and not nv.lib().spmi_gemm_rows_bf16_supported(256, 128)
    gen = torch.Generator().manual_seed(k * n)
    for m in (4096, 5000 + 13):
        x = (torch.randn(m, k, generator=gen) * 2).to(torch.bfloat16).to(dev)
        w = (torch.randn(n, k, generator=gen) * 0.1).to(dev)
        b = torch.randn(n, generator=gen).to(dev)
        rs = (torch.rand(m, generator=gen) + 0.5).to(dev)
        pieces = ops.split_bf16x3(w)
        got = ops.gemm_rows_bf16(x, pieces, n, b, "leaky_relu", 0.01, rs, out_dtype)
        pre = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
        want = torch.where(pre > 0, pre, pre * 0.01) * rs.double().cpu()[:, None]
        assert got.dtype == out_dtype and got.shape == (m, n)
        err = (got.double().cpu() - want).abs()
        if out_dtype == torch.float32:
            assert float(err.max()) < 1e-5 * float(want.abs().max())
        else:
            # one rounding to bfloat16 on top of the float32 sum's own 1e-5 of the scale
            assert bool((err <= want.abs() * 2.0 ** -8 + 1e-5 * float(want.abs().max())).all())


def test_strided_split_and_largest_magnitude_in_one_launch(dev):
    """Planes of a transposed view read through its strides equal the planes of a contiguous
    copy bit for bit, and the same launch leaves the bit pattern of max |w| (NaN on top) in
    the caller's zeroed slot."""
    gen = torch.Generator().manual_seed(77)
    w = (torch.randn(96, 160, generator=gen) * 3).to(dev)      # [in, out] as GraphConv keeps it
    slot = torch.zeros(1, dtype=torch.int32, device=dev)
    got = ops.split_f16x3(w.t(), slot)
    want = ops.split_f16x3(w.t().contiguous())
    assert got.shape == want.shape == (96 // 16, 3, 2, 160, 8)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert float(slot.view(torch.float32)) == float(w.abs().max())
    wide = torch.randn(64, 200, generator=gen).to(dev)
    view = wide[:, 8:8 + 128]                                    # row stride 200, offset 8
    assert torch.equal(ops.split_f16x3(view).view(torch.int16),
                       ops.split_f16x3(view.contiguous()).view(torch.int16))
    w[5, 7] = float("nan")
    slot.zero_()
    ops.split_f16x3(w.t(), slot)
    assert torch.isnan(slot.view(torch.float32)).item()

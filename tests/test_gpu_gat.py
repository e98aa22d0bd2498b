"""GATConv / GAT on the GPU against a torch restatement of DGL 0.8's GATConv.forward in
index ops (parity unpinned by the reference: DGL is absent and it holds no fixture for this
layer).  fp32, 1e-5 relative to the largest magnitude for outputs, attention and gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc
from spiral_amd import _native as nv
from spiral_amd import graph as G
from spiral_amd import ops
from spiral_amd.core.errors import GraphError
from spiral_amd.core.types import GATConv
from spiral_amd.dtdg.models.encoder.implicitTimeEncoder.staticGraphEncoder import GAT
from spiral_amd.dtdg.types import Snapshot
from spiral_amd.graph import SnapshotGraph

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _graph(dev, n, e, seed, k_dst=3.0):
    src, dst = orc.powerlaw_snapshot(n, e, t=seed, k_dst=k_dst)
    return src, dst, SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)


def _reference(conv, src, dst, n, x, act, residual=False):
    cpu = lambda p: None if p is None else p.detach().cpu().clone().requires_grad_()
    params = dict(fc=cpu(conv.fc.weight), al=cpu(conv.attn_l), ar=cpu(conv.attn_r), b=cpu(conv.bias))
    res_w = cpu(conv.res_fc.weight) if isinstance(conv.res_fc, torch.nn.Linear) else None
    xr = x.clone().requires_grad_()
    out, attn = orc.gatconv_torch(src, dst, n, xr, params["fc"], params["al"], params["ar"], params["b"],
                                  conv._num_heads, conv._negative_slope, act, 0.01, res_w, residual)
    return out, attn, xr, params, res_w


@pytest.mark.parametrize("fi,d,heads,act,k_dst", [
    (16, 32, 1, "leaky_relu", 3.0),   # the shape of the reference's examples: one head
    (24, 16, 4, None, 3.0),           # four heads inside one 64-lane tile
    (12, 64, 2, "relu", 6.0),         # one tile per head, hub rows above the threshold
    (8, 128, 1, None, 3.0),           # two tiles per head
    (8, 8, 3, "relu", 3.0),           # heads * width not a multiple of 64
])
def test_gatconv_forward_backward(dev, fi, d, heads, act, k_dst):
    n, e = 1500, 30_000
    src, dst, g = _graph(dev, n, e, seed=fi + d, k_dst=k_dst)
    assert g.csr().n_heavy > 0 or k_dst < 5   # the skewed case must exercise the hub-row kernels
    torch.manual_seed(d)
    fn = {"leaky_relu": F.leaky_relu, "relu": F.relu, None: None}[act]
    conv = GATConv(fi, d, heads, activation=fn, allow_zero_in_degree=True).to(dev)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, fi)
    y_ref, a_ref, xr, pr, _ = _reference(conv, src, dst, n, x, act)
    wgt = torch.randn(n, heads, d)
    (y_ref * wgt).sum().backward()
    xg = x.to(dev).requires_grad_()
    y, attn = conv(g, xg, get_attention=True)
    (y * wgt.to(dev)).sum().backward()
    assert y.shape == (n, heads, d) and attn.shape == (e, heads, 1)
    assert rel(y.detach().cpu(), y_ref.detach()) < 1e-5
    assert rel(attn.cpu(), a_ref.detach()) < 1e-5
    assert rel(xg.grad.cpu(), xr.grad) < 1e-5
    assert rel(conv.fc.weight.grad.cpu(), pr["fc"].grad) < 1e-5
    assert rel(conv.attn_l.grad.cpu(), pr["al"].grad) < 1e-5
    assert rel(conv.attn_r.grad.cpu(), pr["ar"].grad) < 1e-5
    assert rel(conv.bias.grad.cpu(), pr["b"].grad) < 1e-5


@pytest.mark.parametrize("fi,d,heads", [(32, 16, 2), (20, 16, 2)])
def test_gatconv_residual(dev, fi, d, heads):
    n, e = 600, 5000
    src, dst, g = _graph(dev, n, e, seed=3)
    torch.manual_seed(1)
    conv = GATConv(fi, d, heads, residual=True, activation=F.relu, allow_zero_in_degree=True).to(dev)
    assert isinstance(conv.res_fc, torch.nn.Identity) == (fi == d * heads)
    x = torch.randn(n, fi)
    y_ref, _, xr, pr, res_w = _reference(conv, src, dst, n, x, "relu", residual=True)
    y_ref.pow(2).sum().backward()
    xg = x.to(dev).requires_grad_()
    y = conv(g, xg)
    y.pow(2).sum().backward()
    assert rel(y.detach().cpu(), y_ref.detach()) < 1e-5
    assert rel(xg.grad.cpu(), xr.grad) < 1e-5
    assert rel(conv.fc.weight.grad.cpu(), pr["fc"].grad) < 1e-5
    if res_w is not None:
        assert rel(conv.res_fc.weight.grad.cpu(), res_w.grad) < 1e-5


def test_zero_in_degree_and_unsupported_widths(dev):
    n = 50
    src = np.array([0, 1, 2, 2], dtype=np.int64)
    dst = np.array([1, 2, 3, 3], dtype=np.int64)   # node 0 and nodes >= 4 have no in-edge
    g = SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
    x = torch.randn(n, 6, device=dev)
    with pytest.raises(GraphError):
        GATConv(6, 8, 1).to(dev)(g, x)
    conv = GATConv(6, 8, 1, allow_zero_in_degree=True).to(dev)
    with torch.no_grad():
        conv.bias.fill_(0.5)
        y = conv(g, x)
    assert torch.equal(y[0], torch.full((1, 8), 0.5, device=dev))  # no messages: bias only
    with pytest.raises(ValueError):
        GATConv(6, 8, 1, attn_drop=1.0)
    # the backward needs a head width the dot-product kernel can split into lane groups
    conv = GATConv(6, 10, 1, allow_zero_in_degree=True).to(dev)
    with torch.no_grad():
        conv(g, x)                      # forward is fine for any width
    with pytest.raises(nv.SpmiError):
        conv(g, x).sum().backward()


def test_edge_kernels_against_numpy(dev):
    """The three edge kernels on their own, hub rows included, against plain numpy."""
    n, e, heads, d = 3000, 120_000, 2, 32
    src, dst = orc.powerlaw_snapshot(n, e, t=5, k_dst=6.0)
    csr = G.build_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n)
    assert csr.n_heavy > 0
    indptr, indices, _ = orc.coo_to_csr(dst, src, n)
    rng = np.random.default_rng(0)
    el = rng.standard_normal((n, heads)).astype(np.float32)
    er = rng.standard_normal((n, heads)).astype(np.float32)
    attn, raw = ops.edge_softmax_scores(csr, torch.from_numpy(el).to(dev), torch.from_numpy(er).to(dev),
                                        0.2, True)
    rows = np.repeat(np.arange(n), np.diff(indptr))
    raw_ref = (el[indices] + er[rows]).T                                  # [H, E]
    s = np.where(raw_ref > 0, raw_ref, 0.2 * raw_ref).astype(np.float64)
    m = np.full((heads, n), -np.inf)
    np.maximum.at(m, (np.arange(heads)[:, None], rows[None, :]), s)
    p = np.exp(s - m[:, rows])
    den = np.zeros((heads, n))
    np.add.at(den, (np.arange(heads)[:, None], rows[None, :]), p)
    a_ref = p / den[:, rows]
    np.testing.assert_array_equal(raw.cpu().numpy(), raw_ref)
    assert rel(attn.cpu().numpy(), a_ref) < 1e-5
    # rows sum to one per head
    sums = np.zeros((heads, n))
    np.add.at(sums, (np.arange(heads)[:, None], rows[None, :]), attn.cpu().numpy().astype(np.float64))
    assert np.abs(sums[:, np.diff(indptr) > 0] - 1).max() < 1e-5
    # per-edge dot products
    a_mat = rng.standard_normal((n, heads * d)).astype(np.float32)
    b_mat = rng.standard_normal((n, heads * d)).astype(np.float32)
    out = torch.empty(heads, e, device=dev)
    a_dev, b_dev = torch.from_numpy(a_mat).to(dev), torch.from_numpy(b_mat).to(dev)
    nv.check(nv.lib().spmi_sddmm_dot(nv.ptr(csr.indptr), nv.ptr(csr.indices), n, e, nv.ptr(a_dev),
                                     heads * d, nv.ptr(b_dev), heads * d, heads, d, nv.ptr(out),
                                     nv.stream()), "spmi_sddmm_dot")
    want = (a_mat[rows].reshape(e, heads, d).astype(np.float64) *
            b_mat[indices].reshape(e, heads, d)).sum(-1).T
    assert rel(out.cpu().numpy(), want) < 1e-5
    # segment sums
    vals = rng.standard_normal((heads, e)).astype(np.float32)
    seg = torch.empty(n, heads, device=dev)
    vals_dev = torch.from_numpy(vals).to(dev)
    nv.check(nv.lib().spmi_segment_sum(nv.ptr(csr.indptr), n, e, nv.ptr(vals_dev),
                                       heads, nv.ptr(seg), nv.ptr(csr.heavy_rows), csr.n_heavy,
                                       csr.heavy_thresh, nv.stream()), "spmi_segment_sum")
    want = np.zeros((n, heads))
    np.add.at(want, (rows[:, None], np.arange(heads)[None, :]), vals.T.astype(np.float64))
    assert rel(seg.cpu().numpy(), want) < 1e-5


def test_gat_encoder_adds_self_loops(dev):
    n, e, fi, hid = 400, 3000, 10, 16
    src, dst, g = _graph(dev, n, e, seed=9)
    torch.manual_seed(3)
    enc = GAT([2, 1], fi, hid, activation=F.leaky_relu).to(dev)   # zero in-degree check stays on
    x = torch.randn(n, fi)
    g.ndata["h"] = x.to(dev)
    ids = torch.arange(n, device=dev)
    h, ids_out = enc((Snapshot(g, 0), ids))
    assert h.shape == (n, hid) and ids_out is ids
    loops = np.arange(n, dtype=np.int64)
    s2, d2 = np.concatenate([src, loops]), np.concatenate([dst, loops])
    ref = x
    for conv, heads in zip(enc.layers, [2, 1]):
        cpu = lambda p: p.detach().cpu()
        ref, _ = orc.gatconv_torch(s2, d2, n, ref, cpu(conv.fc.weight), cpu(conv.attn_l), cpu(conv.attn_r),
                                   cpu(conv.bias), heads, 0.2, "leaky_relu")
        ref = ref.reshape(n, heads * hid)
    assert rel(h.detach().cpu(), ref.detach()) < 1e-5


def test_attention_dropout(dev):
    """dgl GATConv applies `attn_drop` to the softmax weights before the weighted sum
    (reference: GAT passes GATConv keyword arguments through, staticGraphEncoder.py:170-174).
    In eval mode nothing is dropped; in train mode every returned weight is either 0 or the
    undropped weight / (1 - p), the output is the weighted sum with exactly those weights, and
    the gradients equal those of a torch restatement that applies the same realised mask."""
    n, e, fi, d, heads, p = 1200, 20_000, 16, 32, 2, 0.4
    src, dst, g = _graph(dev, n, e, seed=77)
    torch.manual_seed(1)
    conv = GATConv(fi, d, heads, attn_drop=p, allow_zero_in_degree=True).to(dev)
    x = torch.randn(n, fi)
    conv.eval()
    with torch.no_grad():
        y_eval, a_eval = conv(g, x.to(dev), get_attention=True)
    y_ref, a_ref, _, _, _ = _reference(conv, src, dst, n, x, None)
    assert rel(y_eval.cpu(), y_ref.detach()) < 1e-5 and rel(a_eval.cpu(), a_ref.detach()) < 1e-5

    conv.train()
    xg = x.to(dev).requires_grad_()
    y, a = conv(g, xg, get_attention=True)
    a_cpu = a.detach().cpu()
    kept = a_cpu != 0
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 0.02, frac                      # the drop rate asked for
    assert rel(a_cpu[kept], (a_ref.detach() / (1 - p))[kept]) < 1e-5
    wgt = torch.randn(n, heads, d)
    (y * wgt.to(dev)).sum().backward()

    # the same layer in torch index ops with the realised mask
    cpu = lambda t: t.detach().cpu().clone().requires_grad_()
    fc, al, ar, b, xr = cpu(conv.fc.weight), cpu(conv.attn_l), cpu(conv.attn_r), cpu(conv.bias), x.clone().requires_grad_()
    src_t, dst_t = torch.from_numpy(src), torch.from_numpy(dst)
    ft = (xr @ fc.t()).view(-1, heads, d)
    el, er = (ft * al).sum(-1), (ft * ar).sum(-1)
    ee = F.leaky_relu(el[src_t] + er[dst_t], conv._negative_slope)
    m = torch.full((n, heads), -torch.inf).scatter_reduce(0, dst_t[:, None].expand(-1, heads), ee.detach(), "amax")
    pe = torch.exp(ee - m[dst_t])
    den = torch.zeros(n, heads).index_add_(0, dst_t, pe)
    att = pe / den[dst_t] * (kept[:, :, 0].float() / (1 - p))
    want = torch.zeros(n, heads, d).index_add_(0, dst_t, att[:, :, None] * ft[src_t]) + b.view(1, heads, d)
    (want * wgt).sum().backward()
    assert rel(y.detach().cpu(), want.detach()) < 1e-5
    assert rel(xg.grad.cpu(), xr.grad) < 1e-5
    assert rel(conv.fc.weight.grad.cpu(), fc.grad) < 1e-5
    assert rel(conv.attn_l.grad.cpu(), al.grad) < 1e-5 and rel(conv.attn_r.grad.cpu(), ar.grad) < 1e-5
    assert rel(conv.bias.grad.cpu(), b.grad) < 1e-5

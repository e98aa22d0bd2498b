"""Shapes of the wide-feature workload (BASELINE configs[4]: 256-d features, 512-d hidden,
heavy degree skew) at sizes the oracle finishes in seconds: GraphConv 256 -> 512 (DGL
aggregates first when in_feats <= out_feats) on a k_dst = 6 power-law snapshot whose hubs take
the chunked path, and a 2-layer GRU / LSTM with hidden width 512, against the oracle at the
fp32 bar (1e-5 relative to the largest entry of the compared tensor)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc
from spiral_amd.core.types import GraphConv
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import GRU, LSTM
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP
from spiral_amd.graph import SnapshotGraph

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("grad", [False, True])
def test_graphconv_256_to_512_heavy_skew(dev, grad):
    n, e, fi, fo = 20_000, 600_000, 256, 512
    rng = np.random.default_rng(8)
    src, dst = orc.powerlaw_snapshot(n, e, t=31, k_dst=6.0)
    x = rng.standard_normal((n, fi)).astype(np.float32)
    w = (rng.standard_normal((fi, fo)) * 0.1).astype(np.float32)
    b = rng.standard_normal(fo).astype(np.float32)
    graph = SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
    assert graph.csr().n_heavy > 0
    conv = GraphConv(fi, fo, activation=F.leaky_relu, allow_zero_in_degree=True).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
        conv.bias.copy_(torch.from_numpy(b))
    if not grad:
        want = orc.graphconv_forward(src, dst, n, x, w, b, act="leaky_relu", allow_zero_in_degree=True)
        with torch.no_grad():
            got = conv(graph, torch.from_numpy(x).to(dev))
        assert rel(got.cpu(), want) < 1e-5
        return
    xt, wt, bt = (torch.tensor(a, requires_grad=True) for a in (x, w, b))
    y_ref = orc.graphconv_torch(src, dst, n, xt, wt, bt, act="leaky_relu")
    (y_ref ** 2).mean().backward()
    xg = torch.tensor(x, device=dev, requires_grad=True)
    y = conv(graph, xg)
    (y ** 2).mean().backward()
    assert rel(y.detach().cpu(), y_ref.detach()) < 1e-5
    assert rel(xg.grad.cpu(), xt.grad) < 1e-5
    assert rel(conv.weight.grad.cpu(), wt.grad) < 1e-5
    assert rel(conv.bias.grad.cpu(), bt.grad) < 1e-5


@pytest.mark.parametrize("kind", ["gru", "lstm"])
def test_two_layer_cells_with_hidden_512(dev, kind):
    n, h = 6000, 512
    torch.manual_seed(6)
    cls = {"gru": GRU, "lstm": LSTM}[kind]
    dec = cls(h, h, n, 2, MLP(h, [16])).to(dev)
    dec.eval()
    x = torch.randn(n, h, device=dev) * 2.0
    cpu = lambda t: t.detach().cpu().numpy()
    bm = dec.base_model
    params = [tuple(cpu(getattr(bm, f"{nm}_l{k}")) for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
              for k in range(2)]
    mem_h = cpu(dec.memory_h.memory)
    if kind == "lstm":
        out, h_new, c_new = orc.recurrent_step(kind, cpu(x), mem_h, params, cpu(dec.memory_c.memory))
    else:
        out, h_new = orc.recurrent_step(kind, cpu(x), mem_h, params)
    with torch.no_grad():
        dec((x, None))
    assert rel(cpu(dec.memory_h.memory), h_new) < 1e-5
    if kind == "lstm":
        assert rel(cpu(dec.memory_c.memory), c_new) < 1e-5

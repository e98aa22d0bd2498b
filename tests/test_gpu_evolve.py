"""EvolveGCN-O (weights evolved by a GRU once per snapshot) and bf16 feature storage on
the GPU against this repository's CPU oracle (the reference has no such code: parity is
defined by the oracle).  fp32: 1e-5 relative for outputs, 1e-4 for gradients that pass
through the weight GRU.  bf16 storage: 2e-2 relative to the largest magnitude (inputs,
projection and layer outputs are each rounded to 8 significant bits)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc
from spiral_amd import ops
from spiral_amd.dtdg.types import Snapshot
from spiral_amd.graph import SnapshotGraph
from spiral_amd.nn.evolve import EvolveGCNO

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("rows,cols", [(12, 8), (64, 32), (128, 128)])
def test_dense_gru_cell_gradients(dev, rows, cols):
    torch.manual_seed(rows)
    ref = torch.nn.GRU(cols, cols, batch_first=True)
    w = torch.randn(rows, cols, requires_grad=True)
    o, _ = ref(w[:, None, :], w[None].contiguous())
    o.pow(2).sum().backward()
    params = [getattr(ref, f"{n}_l0").detach().clone().to(dev).requires_grad_() for n in
              ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    wg = w.detach().clone().to(dev).requires_grad_()
    y = ops.gru_cell_dense(wg, wg, *params)
    y.pow(2).sum().backward()
    assert rel(y.detach().cpu(), o[:, 0, :].detach()) < 1e-5
    assert rel(wg.grad.cpu(), w.grad) < 1e-5  # both the input and the state path
    for p, n in zip(params, ("weight_ih", "weight_hh", "bias_ih", "bias_hh")):
        assert rel(p.grad.cpu(), getattr(ref, f"{n}_l0").grad) < 1e-5, n


def _setup(dev, detach):
    n, e, f, hid, steps = 700, 6000, 16, 32, 3
    snaps = [orc.powerlaw_snapshot(n, e, t=t) for t in range(steps)]
    torch.manual_seed(3)
    model = EvolveGCNO(2, f, hid, activation=F.leaky_relu, allow_zero_in_degree=True, detach_state=detach)
    with torch.no_grad():
        for layer in model.layers:
            layer.bias.normal_()
    x = torch.randn(n, f)
    return n, snaps, model, x


@pytest.mark.parametrize("detach", [True, False])
def test_evolve_gcn_o_matches_oracle(dev, detach):
    n, snaps, model, x = _setup(dev, detach)
    init = [p.detach().clone().requires_grad_() for p in model.initial_weights]
    grus = [copy.deepcopy(ev.gru) for ev in model.evolvers]
    biases = [l.bias.detach().clone().requires_grad_() for l in model.layers]
    outs_ref = orc.evolve_gcn_o_torch(snaps, n, x, init, grus, biases, detach_state=detach)
    sum(o.pow(2).mean() for o in outs_ref).backward()

    model.to(dev)
    xg = x.to(dev)
    outs = []
    for t, (src, dst) in enumerate(snaps):
        g = SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
        g.ndata["h"] = xg
        h, _ = model((Snapshot(g, t), None))
        outs.append(h)
    sum(o.pow(2).mean() for o in outs).backward()
    for o, r in zip(outs, outs_ref):
        assert rel(o.detach().cpu(), r.detach()) < 1e-5
    for p, r in zip(model.initial_weights, init):
        assert rel(p.grad.cpu(), r.grad) < 1e-4
    for ev, r in zip(model.evolvers, grus):
        for (name, p), q in zip(ev.gru.named_parameters(), r.parameters()):
            assert rel(p.grad.cpu(), q.grad) < 1e-4, name
    for l, r in zip(model.layers, biases):
        assert rel(l.bias.grad.cpu(), r.grad) < 1e-4
    assert len(model.current_weights()) == 2
    model.reset_weights()
    assert model.current_weights() is None


def test_evolve_gcn_o_bf16_storage(dev):
    n, snaps, model, x = _setup(dev, True)
    ref = copy.deepcopy(model)
    model.storage_dtype = torch.bfloat16
    model.to(dev)
    ref.to(dev)
    xg = x.to(dev)
    for t, (src, dst) in enumerate(snaps):
        g = SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
        g.ndata["h"] = xg
        with torch.no_grad():
            hb, _ = model((Snapshot(g, t), None))
            hf, _ = ref((Snapshot(g, t), None))
        assert hb.dtype == torch.bfloat16 and hf.dtype == torch.float32
        assert rel(hb.float().cpu(), hf.cpu()) < 2e-2
    # training step in bf16 storage runs and yields finite gradients for the weight GRUs
    g.ndata["h"] = xg
    model.reset_weights()
    h, _ = model((Snapshot(g, 0), None), training=True)
    h.float().pow(2).mean().backward()
    for ev in model.evolvers:
        for p in ev.gru.parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all()

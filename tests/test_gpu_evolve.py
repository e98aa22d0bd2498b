"""EvolveGCN-O (weights evolved by a GRU once per snapshot) and bf16 feature storage on
the GPU against this repository's CPU oracle (the reference has no such code: parity is
defined by the oracle).  fp32: 1e-5 relative for outputs and for every gradient, those that
pass through the weight GRU included.  bf16 storage: the outputs against the oracle evaluated
in float64 on the bf16-rounded inputs, 1e-2 relative to the largest magnitude (the projection
and the layer outputs are rounded to 8 significant bits once each, 2^-9 per rounding; sums
are float32).  The weight GRU fed back into itself for 128 snapshots (the length of BASELINE
configs[3]) stays within 1e-5 of float64 and does not drift."""
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


def _setup(dev, detach, n=700, e=6000, f=16, hid=32, steps=3):
    snaps = [orc.powerlaw_snapshot(n, e, t=t) for t in range(steps)]
    torch.manual_seed(3)
    model = EvolveGCNO(2, f, hid, activation=F.leaky_relu, allow_zero_in_degree=True, detach_state=detach)
    with torch.no_grad():
        for layer in model.layers:
            layer.bias.normal_()
    x = torch.randn(n, f)
    return n, snaps, model, x


@pytest.mark.parametrize("detach,shape", [(True, {}), (False, {}),
                                          # 128 -> 128 on 6000 rows: the projection kernels of the bench
                                          (True, dict(n=6000, e=60_000, f=128, hid=128))])
def test_evolve_gcn_o_matches_oracle(dev, detach, shape):
    n, snaps, model, x = _setup(dev, detach, **shape)
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
    worst = {}
    for i, (p, r) in enumerate(zip(model.initial_weights, init)):
        worst[f"W0[{i}]"] = rel(p.grad.cpu(), r.grad)
    for i, (ev, r) in enumerate(zip(model.evolvers, grus)):
        for (name, p), q in zip(ev.gru.named_parameters(), r.parameters()):
            worst[f"gru[{i}].{name}"] = rel(p.grad.cpu(), q.grad)
    for i, (l, r) in enumerate(zip(model.layers, biases)):
        worst[f"bias[{i}]"] = rel(l.bias.grad.cpu(), r.grad)
    print("largest gradient error", max(worst.values()), max(worst, key=worst.get))
    assert max(worst.values()) < 1e-5, worst
    assert len(model.current_weights()) == 2
    model.reset_weights()
    assert model.current_weights() is None


def _float64_oracle(snaps, n, x64, model):
    init = [p.detach().double().cpu() for p in model.initial_weights]
    grus = [copy.deepcopy(ev.gru).double().cpu() for ev in model.evolvers]
    biases = [l.bias.detach().double().cpu() for l in model.layers]
    with torch.no_grad():
        return orc.evolve_gcn_o_torch(snaps, n, x64, init, grus, biases, detach_state=True)


@pytest.mark.parametrize("shape", [{}, dict(n=6000, e=60_000, f=128, hid=128)])
def test_evolve_gcn_o_bf16_storage(dev, shape):
    """bf16 feature storage against the ORACLE in float64 on the bf16-rounded inputs (not against
    this package's own float32 path).  The second shape (128 -> 128, 6000 rows) runs on
    spmi_gemm_rows_bf16 and the bf16 aggregation, the kernels of the config4 bench."""
    n, snaps, model, x = _setup(dev, True, **shape)
    x_bf16 = x.to(torch.bfloat16)
    want = _float64_oracle(snaps, n, x_bf16.double(), model)
    model.storage_dtype = torch.bfloat16
    model.to(dev)
    xg = x_bf16.to(dev)
    worst = 0.0
    for t, (src, dst) in enumerate(snaps):
        g = SnapshotGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
        g.ndata["h"] = xg
        with torch.no_grad():
            hb, _ = model((Snapshot(g, t), None))
        assert hb.dtype == torch.bfloat16
        worst = max(worst, rel(hb.float().cpu(), want[t]))
    print("bf16 storage against the float64 oracle:", worst)
    assert worst < 1e-2, worst
    # training step in bf16 storage runs and yields finite gradients for the weight GRUs
    g.ndata["h"] = xg
    model.reset_weights()
    h, _ = model((Snapshot(g, 0), None), training=True)
    h.float().pow(2).mean().backward()
    for ev in model.evolvers:
        for p in ev.gru.parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all()


def test_weight_gru_does_not_drift_over_128_snapshots(dev):
    """W_t = GRU(W_{t-1}, W_{t-1}) fed back into itself 128 times (BASELINE configs[3]: 128
    snapshots) in the fp16-plane arithmetic of the cell kernels against float64: the error
    stays below 1e-5 of the largest weight at every step and the last quarter of the sequence
    is no worse than three times the first (the bar of tests/test_gpu_drift.py for the node
    memory)."""
    from spiral_amd.nn.evolve import WeightGRU

    steps, rows, cols = 128, 128, 128
    torch.manual_seed(11)
    cell = WeightGRU(cols)
    ref = copy.deepcopy(cell.gru).double()
    w0 = torch.empty(rows, cols)
    torch.nn.init.xavier_uniform_(w0)
    want, w = [], w0.double()
    with torch.no_grad():
        for _ in range(steps):
            o, _ = ref(w[:, None, :], w[None].contiguous())
            w = o[:, 0, :]
            want.append(w.clone())
    cell.to(dev)
    errs, w = [], w0.to(dev)
    with torch.no_grad():
        for t in range(steps):
            w = cell(w)
            errs.append(float((w.double().cpu() - want[t]).abs().max()) / float(want[t].abs().max()))
    print("weight GRU error over 128 snapshots: first 32", max(errs[:32]), "last 32", max(errs[-32:]))
    assert max(errs) < 1e-5, max(errs)
    assert max(errs[-32:]) < 3 * max(errs[:32]), (max(errs[:32]), max(errs[-32:]))

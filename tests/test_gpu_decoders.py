"""Recurrent decoders on the GPU against goldens produced by the reference's own torch
decoders (tests/golden/make_golden.py): outputs, loss, gradients and node memories over two
consecutive snapshots, then a general id list in inference mode.  fp32, 1e-5 relative to
the largest magnitude, gradients through BatchNorm included."""
import os

import numpy as np
import pytest
import torch

from spiral_amd import ops
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import (
    GRU, LSTM, RNN, NodeMemory)
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CLS = {"gru": GRU, "lstm": LSTM, "rnn": RNN}


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_node_memory_round_trip(dev):
    """tests/dtdg/models/decoder/torch/test_node_memory_torch.py:8-26 on the device."""
    mem = NodeMemory(5, 2, 3, device=dev)
    new = torch.randn(3, 2, 2, device=dev)
    mem.update_memory(new, [2, 3])
    assert torch.equal(mem.get_memory([2, 3]), new)
    old = mem.memory.clone()
    mem.reset_state()
    assert not torch.equal(old, mem.memory) and mem.memory.device.type == "cuda"


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
def test_decoder_matches_reference_goldens(dev, kind):
    d = np.load(os.path.join(GOLD, "decoders.npz"))
    n_nodes, in_d, hid, layers = 23, 12, 8, 2
    dec = CLS[kind](in_d, hid, n_nodes, layers, MLP(hid, [6, 5], dropout=0.0))
    state = {k[len(kind) + 7:]: torch.from_numpy(d[k]) for k in d.files if k.startswith(f"{kind}/param/")}
    dec.load_state_dict(state)
    dec.to(dev)
    dec.memory_h.memory = torch.from_numpy(d[f"{kind}/mem_h0"]).to(dev)
    if kind == "lstm":
        dec.memory_c.memory = torch.from_numpy(d[f"{kind}/mem_c0"]).to(dev)
    dec.train()
    for step, n in enumerate((17, 23)):
        x = torch.tensor(d[f"{kind}/step{step}/x"], device=dev, requires_grad=True)
        y = dec((x, torch.arange(n)))
        loss = (y.squeeze() ** 2).mean()
        dec.zero_grad()
        loss.backward()
        assert rel(y.detach().cpu(), d[f"{kind}/step{step}/y"]) < 1e-5
        assert abs(loss.item() - float(d[f"{kind}/step{step}/loss"])) / float(d[f"{kind}/step{step}/loss"]) < 1e-5
        assert rel(dec.memory_h.memory.cpu(), d[f"{kind}/step{step}/mem_h"]) < 1e-5
        if kind == "lstm":
            assert rel(dec.memory_c.memory.cpu(), d[f"{kind}/step{step}/mem_c"]) < 1e-5
        assert rel(x.grad.cpu(), d[f"{kind}/step{step}/grad_x"]) < 1e-5
        for name, p in dec.named_parameters():
            want = d[f"{kind}/step{step}/grad/{name}"]
            if np.abs(want).max() < 1e-7:  # bias in front of a train-mode BatchNorm: zero up to rounding
                assert p.grad.abs().max().item() < 1e-6
            else:
                assert rel(p.grad.cpu(), want) < 1e-5, name
    dec.eval()
    dec.eval_mode()
    ids = torch.from_numpy(d[f"{kind}/eval/ids"])
    with torch.no_grad():
        y = dec((torch.from_numpy(d[f"{kind}/eval/x_full"]).to(dev), ids))
    assert rel(y.cpu(), d[f"{kind}/eval/y"]) < 1e-5
    assert rel(dec.memory_h.memory.cpu(), d[f"{kind}/eval/mem_h"]) < 1e-5  # untouched in eval mode


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("in_d,hid,m,identity", [(24, 10, 20, False), (24, 32, 20, False),
                                                 (32, 32, 20, False), (64, 64, 300, False), (48, 64, 70, False),
                                                 (128, 128, 333, True), (64, 96, 130, False),
                                                 (32, 256, 129, True)])
def test_cells_against_torch_modules(dev, fused, in_d, hid, m, identity):
    """One step of each cell vs torch.nn.{GRU,LSTM,RNN} on the CPU, rows scattered in the
    memory table (or the identity prefix), including the write-back and all gradients.
    `fused` selects the one-kernel MFMA layer step where the shape allows it (in_d % 16
    == 0 and hid % 32 == 0); otherwise library GEMMs + the gate kernel run."""
    torch.manual_seed(hid + in_d)
    ops.USE_FUSED_CELLS = fused
    n_mem = m + 40
    ids = torch.arange(m) if identity else torch.randperm(n_mem)[:m]
    for kind, mod in (("gru", torch.nn.GRU), ("lstm", torch.nn.LSTM), ("rnn", torch.nn.RNN)):
        ref = mod(in_d, hid, batch_first=True)
        x = torch.randn(m, in_d, requires_grad=True)
        mem_h, mem_c = torch.randn(n_mem, hid), torch.randn(n_mem, hid)
        h0 = mem_h[ids][None]
        if kind == "lstm":
            out, (h1, c1) = ref(x[:, None, :], (h0, mem_c[ids][None]))
        else:
            out, h1 = ref(x[:, None, :], h0)
        out.squeeze(1).pow(2).sum().backward()

        params = [getattr(ref, f"{n}_l0").detach().clone().to(dev).requires_grad_() for n in
                  ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        xg = x.detach().clone().to(dev).requires_grad_()
        gh, gc = mem_h.to(dev).clone(), mem_c.to(dev).clone()
        idg = None if identity else ids.to(dev)
        if kind == "gru":
            y = ops.gru_cell(xg, *params, gh, idg)
        elif kind == "rnn":
            y = ops.rnn_cell(xg, *params, gh, idg)
        else:
            y, c_new = ops.lstm_cell(xg, *params, gh, gc, idg)
            want_c = mem_c.clone()
            want_c[ids] = c1[0].detach()
            assert rel(gc.cpu(), want_c) < 1e-5
        y.pow(2).sum().backward()
        assert rel(y.detach().cpu(), out.squeeze(1).detach()) < 1e-5
        want_h = mem_h.clone()
        want_h[ids] = h1[0].detach()
        assert rel(gh.cpu(), want_h) < 1e-5
        assert rel(xg.grad.cpu(), x.grad) < 1e-5
        for p, n in zip(params, ("weight_ih", "weight_hh", "bias_ih", "bias_hh")):
            assert rel(p.grad.cpu(), getattr(ref, f"{n}_l0").grad) < 1e-5, (kind, n)
    ops.USE_FUSED_CELLS = True

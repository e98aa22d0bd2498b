"""Generates the golden vectors under tests/golden/ (run in the build container only).

Sources of truth
  * decoders.npz: the reference's OWN pure-torch decoders
    (spiro/dtdg/models/decoder/torch/sequentialDecoder/implicitTimeModels.py and
    torch/simpleDecoder.py), imported from /root/reference.  `spiro.dtdg/__init__` imports
    DGL, which is not installed, so a bare `spiro.dtdg` namespace package is seeded first
    and the two pure-torch modules are imported below it.
  * types_fixture.npz: the fixture and the assertions of the reference's
    tests/dtdg/test_types_torch.py:12-14,36-56 and tests/core/test_utils.py:47-53,
    written out as data.
  * graphconv.npz: outputs of THIS repository's CPU oracle (DGL is absent, so the
    reference cannot produce them: parity unpinned by the reference); they freeze the
    oracle so that later edits to it are noticed.

Usage: python tests/golden/make_golden.py   (needs /root/reference; writes *.npz here)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference_decoders():
    os.chdir("/tmp")  # importing spiro writes spiro.log into the cwd
    sys.path.insert(0, REF)
    import spiro  # noqa: F401

    for name in ("spiro.dtdg", "spiro.dtdg.models", "spiro.dtdg.models.decoder",
                 "spiro.dtdg.models.decoder.torch",
                 "spiro.dtdg.models.decoder.torch.sequentialDecoder"):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = mod
    from spiro.dtdg.models.decoder.torch.sequentialDecoder import implicitTimeModels as itm
    from spiro.dtdg.models.decoder.torch import simpleDecoder as sd
    return itm, sd


def decoder_goldens():
    itm, sd = import_reference_decoders()
    out = {}
    n_nodes, in_d, hid, layers = 23, 12, 8, 2
    for kind, cls in (("gru", itm.GRU), ("lstm", itm.LSTM), ("rnn", itm.RNN)):
        torch.manual_seed(1234)
        mlp = sd.MLP(hid, [6, 5], dropout=0.0)
        dec = cls(in_d, hid, n_nodes, layers, mlp)
        dec.train()
        for name, p in dec.state_dict().items():
            out[f"{kind}/param/{name}"] = p.detach().numpy().copy()
        out[f"{kind}/mem_h0"] = dec.memory_h.memory.numpy().copy()
        if kind == "lstm":
            out[f"{kind}/mem_c0"] = dec.memory_c.memory.numpy().copy()
        # two consecutive snapshots: 17 nodes, then 23 (node set grows, ids = arange)
        for step, n in enumerate((17, 23)):
            x = torch.randn(n, in_d, requires_grad=True)
            ids = torch.arange(n)
            y = dec((x, ids))
            loss = (y.squeeze() ** 2).mean()
            dec.zero_grad()
            loss.backward()
            out[f"{kind}/step{step}/x"] = x.detach().numpy().copy()
            out[f"{kind}/step{step}/y"] = y.detach().numpy().copy()
            out[f"{kind}/step{step}/loss"] = np.array(loss.item(), dtype=np.float64)
            out[f"{kind}/step{step}/grad_x"] = x.grad.numpy().copy()
            for name, p in dec.named_parameters():
                out[f"{kind}/step{step}/grad/{name}"] = p.grad.numpy().copy()
            out[f"{kind}/step{step}/mem_h"] = dec.memory_h.memory.numpy().copy()
            if kind == "lstm":
                out[f"{kind}/step{step}/mem_c"] = dec.memory_c.memory.numpy().copy()
        # a general (non-identity) id list and inference mode: memory must stay untouched
        dec.eval()
        dec.eval_mode()
        ids = torch.tensor([5, 0, 19, 7, 3])
        x = torch.randn(5, in_d)
        with torch.no_grad():
            y = dec((torch.cat([x, torch.zeros(18, in_d)])[torch.arange(23)], ids))
        out[f"{kind}/eval/x_full"] = torch.cat([x, torch.zeros(18, in_d)]).numpy().copy()
        out[f"{kind}/eval/ids"] = ids.numpy().copy()
        out[f"{kind}/eval/y"] = y.numpy().copy()
        out[f"{kind}/eval/mem_h"] = dec.memory_h.memory.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "decoders.npz"), **out)


def types_fixture():
    edges = np.array([[0, 0, 1], [0, 0, 2], [1, 3, 0], [2, 4, 1]])
    nodes = np.array([[0, 3], [0, 2], [0, 5], [1, 4], [2, 9]])
    times = np.array([2006, 2007, 2008])
    np.savez(os.path.join(HERE, "types_fixture.npz"), edges=edges, nodes=nodes, times=times,
             expect_t=np.array([0, 1]), expect_num_edges=np.array([2, 3]),
             expect_num_nodes=np.array([3, 4]), expect_citation_sum=np.array([0.0, 2.0]),
             anchor_in=np.array([0, 0, 1, 1, 2, 3, 3, 3, 4, 5]),
             anchor_out=np.array([2, 4, 5, 8, 9, 10]))


def graphconv_goldens():
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc

    out = {}
    rng = np.random.default_rng(7)
    # (a) multi-edges, a self-loop and zero in-degree nodes
    src = np.array([0, 0, 0, 1, 2, 2, 3, 5, 5, 5], dtype=np.int64)
    dst = np.array([1, 1, 2, 2, 2, 0, 0, 1, 1, 1], dtype=np.int64)
    n = 7
    x = rng.standard_normal((n, 6)).astype(np.float32)
    out["small/src"], out["small/dst"], out["small/x"] = src, dst, x
    for fo in (4, 6, 9):
        w = rng.standard_normal((6, fo)).astype(np.float32)
        b = rng.standard_normal(fo).astype(np.float32)
        out[f"small/w{fo}"], out[f"small/b{fo}"] = w, b
        for norm in ("both", "right", "left", "none"):
            out[f"small/y_{norm}_{fo}"] = orc.graphconv_forward(
                src, dst, n, x, w, b, norm=norm, act="leaky_relu", allow_zero_in_degree=True)
    # (b) 1k nodes / 8k edges power law
    n, e, f = 1000, 8000, 16
    src, dst = orc.powerlaw_snapshot(n, e, t=0)
    x = rng.standard_normal((n, f)).astype(np.float32)
    w = rng.standard_normal((f, 8)).astype(np.float32) * 0.3
    b = rng.standard_normal(8).astype(np.float32)
    out["pl/src"], out["pl/dst"], out["pl/x"], out["pl/w"], out["pl/b"] = src, dst, x, w, b
    out["pl/y"] = orc.graphconv_forward(src, dst, n, x, w, b, act="relu", allow_zero_in_degree=True)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    bt = torch.tensor(b, requires_grad=True)
    y = orc.graphconv_torch(src, dst, n, xt, wt, bt, act="relu")
    loss = (y ** 2).mean()
    loss.backward()
    out["pl/loss"] = np.array(loss.item())
    out["pl/grad_x"], out["pl/grad_w"], out["pl/grad_b"] = (xt.grad.numpy(), wt.grad.numpy(),
                                                           bt.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "graphconv.npz"), **out)


if __name__ == "__main__":
    types_fixture()
    graphconv_goldens()
    decoder_goldens()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))

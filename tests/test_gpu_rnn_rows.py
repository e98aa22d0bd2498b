"""The whole-table tanh-RNN layer step (spmi_rnn_rows_fwd, csrc/cell_rows.hip): the whole
weight matrix resident in LDS as two fp16 planes, rows streamed past it, new state written to a
second slab.  Checked against the CPU oracle (1e-5 relative to the largest entry of the expected
state), against spmi_cell_f16x3_fwd, which sums the same products in another order, and through
the decoder, whose node memory trades tables."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from spiral_amd import _native as nv
from spiral_amd import ops
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import RNN
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP

pytestmark = pytest.mark.gpu

cpu = lambda t: t.detach().cpu().numpy()


def _params(h, i, dev, seed, scale=0.1):
    gen = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=gen) * scale
    return [t.to(dev) for t in (mk(h, i), mk(h, h), mk(h), mk(h))]


@pytest.mark.parametrize("m", [1, 31, 32, 33, 1000, 2049, 70001])
@pytest.mark.parametrize("bias", [True, False])
def test_against_oracle_and_the_row_tile_kernel(dev, m, bias):
    gen = torch.Generator().manual_seed(m)
    x = (torch.randn(m, 128, generator=gen) * 1.5).to(dev)
    h0 = torch.randn(m, 128, generator=gen).to(dev)
    w_ih, w_hh, b_ih, b_hh = _params(128, 128, dev, seed=3)
    if not bias:
        b_ih = b_hh = None
    out = torch.full((m, 128), float("nan"), device=dev)
    got = ops.rnn_rows(x, w_ih, w_hh, b_ih, b_hh, h0, out)
    assert got is out
    want = orc.rnn_cell(cpu(x), cpu(h0), cpu(w_ih), cpu(w_hh), cpu(b_ih) if bias else None,
                        cpu(b_hh) if bias else None)
    assert np.abs(cpu(out) - want).max() < 1e-5 * np.abs(want).max()
    with torch.no_grad():
        old, ops.CELL_GEMM = ops.CELL_GEMM, "f16x3"
        try:
            other = ops.rnn_cell(x, w_ih, w_hh, b_ih, b_hh, h0.clone(), None, mem_write=False)
        finally:
            ops.CELL_GEMM = old
    assert (out - other).abs().max().item() < 2e-6      # same products, different summation order


def test_strided_x_big_rows_and_padding(dev):
    """x with a row stride, rows of x beyond fp16's range (split after a power-of-two scaling),
    the slabs' neighbours untouched."""
    m = 555
    gen = torch.Generator().manual_seed(9)
    wide = torch.randn(m, 200, generator=gen)
    wide[7, 11] = 4.0e4
    wide[8, 11], wide[8, 48] = 7.0e4, -6.6e4
    wide[300] *= 3.0e8
    w_ih = torch.randn(128, 128, generator=gen) * 0.1
    small = torch.randint(-4, 5, (128, 2), generator=gen).float() * 2.0 ** -14
    w_ih[:, 3], w_ih[:, 40] = small[:, 0], small[:, 1]      # columns 11 and 48 of `wide`
    w_hh = torch.randn(128, 128, generator=gen) * 0.1
    b_ih, b_hh = torch.randn(128, generator=gen) * 0.1, torch.randn(128, generator=gen) * 0.1
    h0 = torch.randn(m, 128, generator=gen)
    x = wide[:, 8:136]                                      # row stride 200, 16-byte aligned start
    want = torch.tanh(x.double() @ w_ih.double().t() + b_ih.double()
                      + h0.double() @ w_hh.double().t() + b_hh.double()).numpy()
    slab = torch.full((m + 2, 128), 7.0, device=dev)
    ops.rnn_rows(wide.to(dev)[:, 8:136], w_ih.to(dev), w_hh.to(dev), b_ih.to(dev), b_hh.to(dev), h0.to(dev),
                 slab[1:-1])
    assert np.abs(cpu(slab[1:-1]) - want).max() < 1e-5 * np.abs(want).max()
    assert (slab[0] == 7.0).all() and (slab[-1] == 7.0).all()


def test_overlap_and_shapes_are_rejected(dev):
    m = 64
    x = torch.randn(m, 128, device=dev)
    w_ih, w_hh, b_ih, b_hh = _params(128, 128, dev, seed=1)
    mem = torch.randn(m + 10, 128, device=dev)
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.rnn_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], mem[:m])
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.rnn_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], mem[10:])
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.rnn_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], x)
    assert not nv.lib().spmi_rnn_rows_supported(64, 128)
    assert not nv.lib().spmi_rnn_rows_supported(128, 256)


def test_decoder_trades_memory_tables(dev):
    """RNN decoder over every node under no_grad: outputs and memory equal the row-tile path
    (USE_GRU_ROWS off) over three steps, in training mode (memory written) and eval mode
    (memory left alone); a batch of some nodes keeps taking the in-place kernel."""
    n = 3000
    torch.manual_seed(0)
    dec = RNN(128, 128, n, 2, MLP(128, [32, 8])).to(dev)
    dec.eval()                                            # BatchNorm on running statistics
    start = dec.memory_h.memory.clone()
    ids = torch.arange(n, device=dev)
    xs = [torch.randn(n, 128, device=dev) for _ in range(3)]

    def run(flag):
        old, ops.USE_GRU_ROWS = ops.USE_GRU_ROWS, flag
        try:
            dec.memory_h.memory = start.clone()
            dec.memory_h._spare = None
            outs = []
            with torch.no_grad():
                dec.training_mode = True
                for x in xs[:2]:
                    outs.append(dec((x, ids)).clone())
                after_train = dec.memory_h.memory.clone()
                dec.training_mode = False
                outs.append(dec((xs[2], ids)).clone())
                assert torch.equal(dec.memory_h.memory, after_train)
                dec.training_mode = True
                some = torch.arange(0, n, 3, device=dev)
                outs.append(dec((xs[2], some)).clone())
            return outs, after_train, dec.memory_h.memory.clone()
        finally:
            ops.USE_GRU_ROWS = old

    a_out, a_mid, a_end = run(True)
    b_out, b_mid, b_end = run(False)
    assert not torch.equal(a_mid, start)
    for a, b in zip(a_out, b_out):
        assert (a - b).abs().max().item() < 1e-5
    assert (a_mid - b_mid).abs().max().item() < 4e-6
    assert (a_end - b_end).abs().max().item() < 4e-6

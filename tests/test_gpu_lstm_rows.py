"""The whole-table LSTM layer step (spmi_lstm_rows_fwd, csrc/cell_rows.hip): the four gates'
weights of a 32-unit column slice resident in LDS as two fp16 planes, rows streamed past them,
new hidden state written to a second slab, cell state updated in place or into a slab of its
own.  Checked against the CPU oracle (1e-5 relative to the largest entry of the expected state:
the fp32 bar of the other recurrent kernels), against spmi_cell_f16x3_fwd, which sums the same
products in another order, and through the decoder, whose hidden-state memory trades tables."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from spiral_amd import _native as nv
from spiral_amd import ops
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import LSTM
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP

pytestmark = pytest.mark.gpu

cpu = lambda t: t.detach().cpu().numpy()


def _params(h, i, dev, seed, scale=0.1):
    gen = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=gen) * scale
    return [t.to(dev) for t in (mk(4 * h, i), mk(4 * h, h), mk(4 * h), mk(4 * h))]


@pytest.mark.parametrize("m", [1, 31, 32, 33, 1000, 2049, 70001])
@pytest.mark.parametrize("bias", [True, False])
def test_against_oracle_and_the_row_tile_kernel(dev, m, bias):
    gen = torch.Generator().manual_seed(m)
    x = (torch.randn(m, 128, generator=gen) * 1.5).to(dev)
    h0 = torch.randn(m, 128, generator=gen).to(dev)
    c0 = (torch.randn(m, 128, generator=gen) * 2.0).to(dev)
    w_ih, w_hh, b_ih, b_hh = _params(128, 128, dev, seed=3)
    if not bias:
        b_ih = b_hh = None
    h1 = torch.full((m, 128), float("nan"), device=dev)
    c1 = torch.full((m, 128), float("nan"), device=dev)
    got_h, got_c = ops.lstm_rows(x, w_ih, w_hh, b_ih, b_hh, h0, c0, h1, c1)
    assert got_h is h1 and got_c is c1
    want_h, want_c = orc.lstm_cell(cpu(x), cpu(h0), cpu(c0), cpu(w_ih), cpu(w_hh),
                                   cpu(b_ih) if bias else None, cpu(b_hh) if bias else None)
    assert np.abs(cpu(h1) - want_h).max() < 1e-5 * np.abs(want_h).max()
    assert np.abs(cpu(c1) - want_c).max() < 1e-5 * np.abs(want_c).max()
    with torch.no_grad():
        old, ops.CELL_GEMM = ops.CELL_GEMM, "f16x3"
        try:
            oh, oc = ops.lstm_cell(x, w_ih, w_hh, b_ih, b_hh, h0.clone(), c0.clone(), None, mem_write=False)
        finally:
            ops.CELL_GEMM = old
    assert (h1 - oh).abs().max().item() < 2e-6           # same products, different summation order
    assert (c1 - oc).abs().max().item() < 4e-6
    # the cell state in place: same values
    c_inplace = c0.clone()
    h2 = torch.empty_like(h1)
    ops.lstm_rows(x, w_ih, w_hh, b_ih, b_hh, h0, c_inplace, h2, c_inplace)
    assert torch.equal(h2, h1) and torch.equal(c_inplace, c1)


def test_strided_x_and_padding_untouched(dev):
    m = 555
    gen = torch.Generator().manual_seed(9)
    wide = torch.randn(m, 200, generator=gen).to(dev)
    x = wide[:, 8:136]                                   # row stride 200, 16-byte aligned start
    h0 = torch.randn(m, 128, generator=gen).to(dev)
    c0 = torch.randn(m, 128, generator=gen).to(dev)
    w_ih, w_hh, b_ih, b_hh = _params(128, 128, dev, seed=5)
    hs = torch.full((m + 2, 128), 7.0, device=dev)
    cs = torch.full((m + 2, 128), 7.0, device=dev)
    ops.lstm_rows(x, w_ih, w_hh, b_ih, b_hh, h0, c0, hs[1:-1], cs[1:-1])
    want_h, want_c = orc.lstm_cell(cpu(x), cpu(h0), cpu(c0), cpu(w_ih), cpu(w_hh), cpu(b_ih), cpu(b_hh))
    assert np.abs(cpu(hs[1:-1]) - want_h).max() < 1e-5 * np.abs(want_h).max()
    assert np.abs(cpu(cs[1:-1]) - want_c).max() < 1e-5 * np.abs(want_c).max()
    for slab in (hs, cs):
        assert (slab[0] == 7.0).all() and (slab[-1] == 7.0).all()


def test_rows_of_x_of_any_magnitude(dev):
    """Rows that reach 2^15 are split after a multiplication by a power of two and their partial
    sums scaled back before the state terms."""
    m = 300
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(m, 128, generator=gen)
    w_ih = torch.randn(512, 128, generator=gen) * 0.1
    small = torch.randint(-4, 5, (512, 2), generator=gen).float() * 2.0 ** -14
    w_ih[:, 3], w_ih[:, 40] = small[:, 0], small[:, 1]
    x[7, 3] = 4.0e4
    x[8, 3], x[8, 40] = 7.0e4, -6.6e4
    x[9, 40] = 1.0e5
    x[200, 3] = -9.0e4
    x[299] *= 3.0e8
    w_hh = torch.randn(512, 128, generator=gen) * 0.1
    b_ih, b_hh = torch.randn(512, generator=gen) * 0.1, torch.randn(512, generator=gen) * 0.1
    h0, c0 = torch.randn(m, 128, generator=gen), torch.randn(m, 128, generator=gen)
    g = x.double() @ w_ih.double().t() + b_ih.double() + h0.double() @ w_hh.double().t() + b_hh.double()
    i, f = torch.sigmoid(g[:, :128]), torch.sigmoid(g[:, 128:256])
    gg, o = torch.tanh(g[:, 256:384]), torch.sigmoid(g[:, 384:])
    want_c = f * c0.double() + i * gg
    want_h = (o * torch.tanh(want_c)).numpy()
    want_c = want_c.numpy()
    args = [t.to(dev) for t in (x, w_ih, w_hh, b_ih, b_hh, h0, c0)]
    h1, c1 = ops.lstm_rows(*args, torch.empty(m, 128, device=dev), torch.empty(m, 128, device=dev))
    assert np.abs(cpu(h1) - want_h).max() < 1e-5 * np.abs(want_h).max()
    assert np.abs(cpu(c1) - want_c).max() < 1e-5 * np.abs(want_c).max()
    # the promise of bounded rows is a hint this kernel does not need: same results with it
    h2, c2 = ops.lstm_rows(*args, torch.empty(m, 128, device=dev), torch.empty(m, 128, device=dev),
                           x_bounded=True)
    assert torch.equal(h2, h1) and torch.equal(c2, c1)


def test_overlap_and_shapes_are_rejected(dev):
    m = 64
    x = torch.randn(m, 128, device=dev)
    w_ih, w_hh, b_ih, b_hh = _params(128, 128, dev, seed=1)
    mem = torch.randn(m + 10, 128, device=dev)
    c = torch.randn(m, 128, device=dev)
    spare = torch.empty(m, 128, device=dev)
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.lstm_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], c, mem[:m], spare)
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.lstm_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], c, mem[10:], spare)
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.lstm_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], c, x, spare)
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.lstm_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], c, spare, mem[5:m + 5])
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.lstm_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], c, c, spare)
    assert not nv.lib().spmi_lstm_rows_supported(64, 128)
    assert not nv.lib().spmi_lstm_rows_supported(128, 256)


def test_decoder_trades_memory_tables(dev):
    """LSTM decoder over every node under no_grad: outputs and both memories equal the row-tile
    path (USE_GRU_ROWS off) over three steps, in training mode (memories written) and eval mode
    (memories left alone); a batch of some nodes keeps taking the in-place kernel."""
    n = 3000
    torch.manual_seed(0)
    dec = LSTM(128, 128, n, 2, MLP(128, [32, 8])).to(dev)
    dec.eval()                                            # BatchNorm on running statistics
    start_h, start_c = dec.memory_h.memory.clone(), dec.memory_c.memory.clone()
    ids = torch.arange(n, device=dev)
    xs = [torch.randn(n, 128, device=dev) for _ in range(3)]

    def run(flag):
        old, ops.USE_GRU_ROWS = ops.USE_GRU_ROWS, flag
        try:
            dec.memory_h.memory, dec.memory_c.memory = start_h.clone(), start_c.clone()
            dec.memory_h._spare = None
            outs = []
            with torch.no_grad():
                dec.training_mode = True
                for x in xs[:2]:
                    outs.append(dec((x, ids)).clone())
                mid = dec.memory_h.memory.clone(), dec.memory_c.memory.clone()
                dec.training_mode = False
                outs.append(dec((xs[2], ids)).clone())
                assert torch.equal(dec.memory_h.memory, mid[0]) and torch.equal(dec.memory_c.memory, mid[1])
                dec.training_mode = True
                some = torch.arange(0, n, 3, device=dev)
                outs.append(dec((xs[2], some)).clone())
            return outs, mid, (dec.memory_h.memory.clone(), dec.memory_c.memory.clone())
        finally:
            ops.USE_GRU_ROWS = old

    a_out, a_mid, a_end = run(True)
    b_out, b_mid, b_end = run(False)
    assert not torch.equal(a_mid[0], start_h) and not torch.equal(a_mid[1], start_c)
    for a, b in zip(a_out, b_out):
        assert (a - b).abs().max().item() < 1e-5
    for a, b in zip(a_mid + a_end, b_mid + b_end):
        assert (a - b).abs().max().item() < 8e-6

"""The whole-table GRU layer step (spmi_gru_rows_fwd, csrc/cell_rows.hip): gate weights of a
32-unit column slice resident in LDS, rows streamed past them, new state written to a second
slab.  Checked against the CPU oracle (1e-5 relative to the largest entry of the expected state: the fp32 bar of
the other recurrent kernels), against spmi_cell_f16x3_fwd, which computes the same sums in a
different instruction order, and through the decoder, whose node memory trades tables."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from spiral_amd import _native as nv
from spiral_amd import ops
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import GRU
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP

pytestmark = pytest.mark.gpu

cpu = lambda t: t.detach().cpu().numpy()


def _params(h, i, dev, seed, scale=0.1):
    gen = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=gen) * scale
    return [t.to(dev) for t in (mk(3 * h, i), mk(3 * h, h), mk(3 * h), mk(3 * h))]


@pytest.mark.parametrize("m", [1, 31, 32, 33, 1000, 2048, 2049, 70001])
@pytest.mark.parametrize("bias", [True, False])
def test_against_oracle_and_the_row_tile_kernel(dev, m, bias):
    gen = torch.Generator().manual_seed(m)
    x = (torch.randn(m, 128, generator=gen) * 1.5).to(dev)
    h0 = torch.randn(m, 128, generator=gen).to(dev)
    w_ih, w_hh, b_ih, b_hh = _params(128, 128, dev, seed=3)
    if not bias:
        b_ih = b_hh = None
    out = torch.full((m, 128), float("nan"), device=dev)
    got = ops.gru_rows(x, w_ih, w_hh, b_ih, b_hh, h0, out)
    assert got is out
    zeros = np.zeros(3 * 128, np.float32)
    want = orc.gru_cell(cpu(x), cpu(h0), cpu(w_ih), cpu(w_hh), cpu(b_ih) if bias else zeros,
                        cpu(b_hh) if bias else zeros)
    assert np.abs(cpu(out) - want).max() < 1e-5 * np.abs(want).max()
    with torch.no_grad():
        old, ops.CELL_GEMM = ops.CELL_GEMM, "f16x3"
        try:
            other = ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, h0.clone(), None, mem_write=False)
        finally:
            ops.CELL_GEMM = old
    assert (out - other).abs().max().item() < 2e-6      # same products, different summation order


def test_strided_x_and_padding_untouched(dev):
    m = 555
    gen = torch.Generator().manual_seed(9)
    wide = torch.randn(m, 200, generator=gen).to(dev)
    x = wide[:, 8:136]                                   # row stride 200, 16-byte aligned start
    h0 = torch.randn(m, 128, generator=gen).to(dev)
    w_ih, w_hh, b_ih, b_hh = _params(128, 128, dev, seed=5)
    slab = torch.full((m + 2, 128), 7.0, device=dev)
    ops.gru_rows(x, w_ih, w_hh, b_ih, b_hh, h0, slab[1:-1])
    want = orc.gru_cell(cpu(x), cpu(h0), cpu(w_ih), cpu(w_hh), cpu(b_ih), cpu(b_hh))
    assert np.abs(cpu(slab[1:-1]) - want).max() < 1e-5 * np.abs(want).max()
    assert (slab[0] == 7.0).all() and (slab[-1] == 7.0).all()


def test_rows_of_x_of_any_magnitude(dev):
    """Rows that reach 2^15 are scaled by a power of two before the fp16 split and their partial
    sums scaled back before the state terms; a wrong promise of bounded rows fails loudly."""
    m = 300
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(m, 128, generator=gen)
    w_ih = torch.randn(384, 128, generator=gen) * 0.1
    small = torch.randint(-4, 5, (384, 2), generator=gen).float() * 2.0 ** -14
    w_ih[:, 3], w_ih[:, 40] = small[:, 0], small[:, 1]
    big_rows = (7, 8, 9, 200, 299)
    x[7, 3] = 4.0e4
    x[8, 3], x[8, 40] = 7.0e4, -6.6e4
    x[9, 40] = 1.0e5
    x[200, 3] = -9.0e4
    x[299] *= 3.0e8
    w_hh = torch.randn(384, 128, generator=gen) * 0.1
    b_ih, b_hh = torch.randn(384, generator=gen) * 0.1, torch.randn(384, generator=gen) * 0.1
    h0 = torch.randn(m, 128, generator=gen)
    gi = x.double() @ w_ih.double().t() + b_ih.double()
    gh = h0.double() @ w_hh.double().t() + b_hh.double()
    r = torch.sigmoid(gi[:, :128] + gh[:, :128])
    z = torch.sigmoid(gi[:, 128:256] + gh[:, 128:256])
    n = torch.tanh(gi[:, 256:] + r * gh[:, 256:])
    want = (n + z * (h0.double() - n)).numpy()
    args = [t.to(dev) for t in (x, w_ih, w_hh, b_ih, b_hh, h0)]
    out = ops.gru_rows(*args, torch.empty(m, 128, device=dev))
    assert np.abs(cpu(out) - want).max() < 1e-5 * np.abs(want).max()
    bad = cpu(ops.gru_rows(*args, torch.empty(m, 128, device=dev), x_bounded=True))
    for row in (8, 9, 200, 299):                         # past fp16's range: inf planes
        assert not np.isfinite(bad[row]).all()
    rest = np.setdiff1d(np.arange(m), big_rows)
    assert np.abs(bad[rest] - want[rest]).max() < 1e-5 * np.abs(want[rest]).max()


def test_overlap_and_shapes_are_rejected(dev):
    m = 64
    x = torch.randn(m, 128, device=dev)
    w_ih, w_hh, b_ih, b_hh = _params(128, 128, dev, seed=1)
    mem = torch.randn(m + 10, 128, device=dev)
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.gru_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], mem[:m])
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.gru_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], mem[10:])
    with pytest.raises(nv.SpmiError, match="overlaps"):
        ops.gru_rows(x, w_ih, w_hh, b_ih, b_hh, mem[:m], x)
    assert not nv.lib().spmi_gru_rows_supported(64, 128)
    assert not nv.lib().spmi_gru_rows_supported(128, 256)


def test_decoder_trades_memory_tables(dev):
    """GRU decoder over every node under no_grad: outputs and memories equal the row-tile path
    (USE_GRU_ROWS off) over three steps, in training mode (memory written) and eval mode
    (memory left alone); a batch of some nodes keeps taking the in-place kernel."""
    n = 3000
    torch.manual_seed(0)
    dec = GRU(128, 128, n, 2, MLP(128, [32, 8])).to(dev)
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

"""The one-kernel recurrent layer step in its three arithmetic modes, against the CPU
oracle: "fp32" (v_mfma_f32_32x32x2_f32), "bf16x3" (fp32 operands split exactly into three
bf16 pieces, six piece products per term on the bf16 matrix cores) and "f16x3" (planes
fp16(v), fp16(v/64), fp16((v - fp16(v))*64), three products per term).  All must meet the same fp32
bar (1e-5 relative to the largest entry of the expected state)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from spiral_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f16x3", "bf16x3", "fp32"])
def mode(request):
    old = ops.CELL_GEMM
    ops.CELL_GEMM = request.param
    yield request.param
    ops.CELL_GEMM = old


def _params(g, i, h, dev, seed):
    gen = torch.Generator().manual_seed(seed)
    mk = lambda *s: (torch.randn(*s, generator=gen) * 0.1)  # about the scale of torch's uniform(-1/sqrt(H), 1/sqrt(H))
    return [t.to(dev) for t in (mk(g * h, i), mk(g * h, h), mk(g * h), mk(g * h))]


@pytest.mark.parametrize("m,i,h,gather", [(1000, 128, 128, False), (777, 64, 32, True), (130, 32, 96, False),
                                          (4097, 160, 64, True), (129, 256, 256, False)])
def test_gru_lstm_rnn_steps(dev, mode, m, i, h, gather):
    cpu = lambda t: t.detach().cpu().numpy()
    n_mem = m + 50 if gather else m
    gen = torch.Generator().manual_seed(m + i + h)
    x = (torch.randn(m, i, generator=gen) * 1.5).to(dev)
    ids = torch.randperm(n_mem, generator=gen)[:m].to(dev) if gather else None
    sel = cpu(ids) if gather else np.arange(m)
    for kind, g in (("gru", 3), ("lstm", 4), ("rnn", 1)):
        w_ih, w_hh, b_ih, b_hh = _params(g, i, h, dev, seed=g)
        mem_h = torch.randn(n_mem, h, generator=gen).to(dev)
        mem_c = torch.randn(n_mem, h, generator=gen).to(dev)
        h0, c0 = cpu(mem_h).copy(), cpu(mem_c).copy()
        with torch.no_grad():
            if kind == "gru":
                out = ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mem_h, ids)
                want = orc.gru_cell(cpu(x), h0[sel], cpu(w_ih), cpu(w_hh), cpu(b_ih), cpu(b_hh))
            elif kind == "lstm":
                out = ops.lstm_cell(x, w_ih, w_hh, b_ih, b_hh, mem_h, mem_c, ids)
                out = out[0] if isinstance(out, tuple) else out
                want, want_c = orc.lstm_cell(cpu(x), h0[sel], c0[sel], cpu(w_ih), cpu(w_hh), cpu(b_ih), cpu(b_hh))
                assert np.abs(cpu(mem_c)[sel] - want_c).max() < 1e-5 * np.abs(want_c).max()
            else:
                out = ops.rnn_cell(x, w_ih, w_hh, b_ih, b_hh, mem_h, ids)
                want = orc.rnn_cell(cpu(x), h0[sel], cpu(w_ih), cpu(w_hh), cpu(b_ih), cpu(b_hh))
        assert np.abs(cpu(out) - want).max() < 1e-5 * np.abs(want).max(), (kind, mode)
        after = cpu(mem_h)
        assert np.array_equal(after[sel], cpu(out))                 # rows written back in place
        untouched = np.setdiff1d(np.arange(n_mem), sel)
        assert np.array_equal(after[untouched], h0[untouched])      # and only those rows


def test_split_pieces_are_exact(dev):
    w = torch.randn(96, 64, device=dev) * torch.logspace(-6, 3, 64, device=dev)
    pieces = ops.split_bf16x3(w)                                    # [cols/16, 3, 2, rows, 8]
    assert pieces.shape == (4, 3, 2, 96, 8) and pieces.dtype == torch.bfloat16
    total = pieces.float().sum(1)                                   # [4, 2, 96, 8]
    back = total.permute(2, 0, 1, 3).reshape(96, 64)
    assert torch.equal(back, w)


def test_f16x3_planes(dev):
    """p0 + p2/64 reproduces w to ~2^-22 and p1 = w/64 to fp16 rounding, over fp16's range."""
    w = torch.randn(96, 64, device=dev) * torch.logspace(-3, 3, 64, device=dev)
    planes = ops.split_f16x3(w)                                     # [cols/16, 3, 2, rows, 8]
    assert planes.shape == (4, 3, 2, 96, 8) and planes.dtype == torch.float16
    un = lambda p: p.double().permute(2, 0, 1, 3).reshape(96, 64)
    p0, p1, p2 = un(planes[:, 0]), un(planes[:, 1]), un(planes[:, 2])
    wd = w.double()
    assert ((p0 + p2 / 64 - wd).abs() <= wd.abs() * 2.0 ** -21 + 2.0 ** -31).all()
    assert ((p1 * 64 - wd).abs() <= wd.abs() * 2.0 ** -11 + 2.0 ** -19).all()


def test_f16x3_rows_of_x_of_any_magnitude(dev):
    """GraphConv outputs of graphs with count-valued features exceed fp16's range.  The default
    kernel scales such rows of x by a power of two (the state is left alone): results must match
    a float64 evaluation, and a caller's wrong promise of bounded input must fail loudly
    (inf/NaN in that row, every other row untouched), not quietly."""
    old, ops.CELL_GEMM = ops.CELL_GEMM, "f16x3"
    try:
        m, i, h = 300, 64, 128
        gen = torch.Generator().manual_seed(4)
        x = torch.randn(m, i, generator=gen)
        w_ih = torch.randn(3 * h, i, generator=gen) * 0.1
        # columns 3 and 40 of the weights: small multiples of 2^-14, so that the outliers below
        # leave the gates out of saturation and the expected value is sensitive to them
        small = torch.randint(-4, 5, (3 * h, 2), generator=gen).float() * 2.0 ** -14
        w_ih[:, 3], w_ih[:, 40] = small[:, 0], small[:, 1]
        big_rows = (7, 8, 9, 200, 299)
        x[7, 3] = 4.0e4                                     # just past 2^15
        x[8, 3], x[8, 40] = 7.0e4, -6.6e4                   # past fp16's largest number
        x[9, 40] = 1.0e5
        x[200, 3] = -9.0e4                                  # second 128-row tile
        x[299] *= 3.0e8                                     # saturates every gate
        w_hh = torch.randn(3 * h, h, generator=gen) * 0.1
        b_ih, b_hh = torch.randn(3 * h, generator=gen) * 0.1, torch.randn(3 * h, generator=gen) * 0.1
        h0 = torch.randn(m, h, generator=gen)
        xd, hd = x.double(), h0.double()
        gi = xd @ w_ih.double().t() + b_ih.double()
        gh = hd @ w_hh.double().t() + b_hh.double()
        r = torch.sigmoid(gi[:, :h] + gh[:, :h])
        z = torch.sigmoid(gi[:, h:2 * h] + gh[:, h:2 * h])
        n = torch.tanh(gi[:, 2 * h:] + r * gh[:, 2 * h:])
        want = (1 - z) * n + z * hd
        to = lambda t: t.to(dev)
        mem = to(h0.clone())
        with torch.no_grad():
            out = ops.gru_cell(to(x), to(w_ih), to(w_hh), to(b_ih), to(b_hh), mem)
        assert torch.isfinite(out).all()
        err = (out.double().cpu() - want).abs()
        assert float(err.max()) < 2e-5, [float(err[r_].max()) for r_ in big_rows]
        assert torch.equal(mem, out)
        mem = to(h0.clone())
        with torch.no_grad():
            bad = ops.gru_cell(to(x), to(w_ih), to(w_hh), to(b_ih), to(b_hh), mem, x_bounded=True)
        for row in (8, 9, 200, 299):                        # beyond 65504: no fp16 value
            assert not torch.isfinite(bad[row]).all()
        keep = torch.ones(m, dtype=torch.bool)
        keep[list(big_rows)] = False
        assert float((bad.double().cpu()[keep] - want[keep]).abs().max()) < 2e-5
        # LSTM and RNN share the staging code; a saturated and an unsaturated big row each
        for cell, g in (("lstm", 4), ("rnn", 1)):
            w1 = torch.randn(g * h, i, generator=gen) * 0.1
            w1[:, 3] = torch.randint(-4, 5, (g * h,), generator=gen).float() * 2.0 ** -14
            w1[:, 40] = 0.0
            w2 = torch.randn(g * h, h, generator=gen) * 0.1
            b1, b2 = torch.randn(g * h, generator=gen) * 0.1, torch.randn(g * h, generator=gen) * 0.1
            pre = xd @ w1.double().t() + b1.double() + hd @ w2.double().t() + b2.double()
            if cell == "rnn":
                want = torch.tanh(pre)
                mem = to(h0.clone())
                with torch.no_grad():
                    out = ops.rnn_cell(to(x), to(w1), to(w2), to(b1), to(b2), mem)
            else:
                c0 = torch.randn(m, h, generator=gen)
                ig, fg = torch.sigmoid(pre[:, :h]), torch.sigmoid(pre[:, h:2 * h])
                gg, og = torch.tanh(pre[:, 2 * h:3 * h]), torch.sigmoid(pre[:, 3 * h:])
                want = og * torch.tanh(fg * c0.double() + ig * gg)
                mem, cmem = to(h0.clone()), to(c0.clone())
                with torch.no_grad():
                    out = ops.lstm_cell(to(x), to(w1), to(w2), to(b1), to(b2), mem, cmem)
                    out = out[0] if isinstance(out, tuple) else out
            assert torch.isfinite(out).all(), cell
            assert float((out.double().cpu() - want).abs().max()) < 2e-5, cell
    finally:
        ops.CELL_GEMM = old


def test_f16x3_out_of_range_state_is_loud(dev):
    """The state has no scaling pass (gate outputs are bounded by 1): a state beyond fp16's
    range must not produce a plausible wrong number."""
    old, ops.CELL_GEMM = ops.CELL_GEMM, "f16x3"
    try:
        w_ih, w_hh, b_ih, b_hh = _params(3, 32, 32, dev, seed=1)
        x = torch.ones(130, 32, device=dev)
        mem = torch.zeros(130, 32, device=dev)
        mem[7, 3] = 1.0e5
        with torch.no_grad():
            out = ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mem)
        assert not torch.isfinite(out[7]).all()
        assert torch.isfinite(out[:7]).all() and torch.isfinite(out[8:]).all()
    finally:
        ops.CELL_GEMM = old


def test_inference_step_keeps_no_gate_values(dev):
    """Under no_grad a cell must not allocate the per-row gate values of its backward pass,
    even though its weights require grad (ctx.needs_input_grad says True regardless)."""
    m, h = 65_536, 128
    w_ih, w_hh, b_ih, b_hh = (t.requires_grad_() for t in _params(3, h, h, dev, seed=9))
    x = torch.randn(m, h, device=dev)
    mem = torch.zeros(m, h, device=dev)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        out = ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mem)
    torch.cuda.synchronize()
    extra = torch.cuda.max_memory_allocated() - base
    assert extra < 2.5 * m * h * 4, extra          # the output and the small weight planes, not 5 H per row
    # and with tracking on, the backward still works through the same entry point
    mem.zero_()
    out = ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mem)
    out.sum().backward()
    assert w_ih.grad is not None and torch.isfinite(w_ih.grad).all()


@pytest.mark.parametrize("m,i,h", [(6000, 128, 128), (5000, 64, 32), (300, 128, 128)])
def test_gru_backward_with_packed_gate_gradients(dev, m, i, h):
    """The GRU layer's backward with the gate gradients as one [rows, 4H] matrix (the default)
    against the two-matrix form and against torch.nn.GRUCell in float64: gradients of x, both
    weights and both biases within 1e-5 of each tensor's largest entry.  At 6000 rows and
    H = 128 dX runs on spmi_gemm_rows with K = 384 and the weight gradients on whole-block
    kernels reading column ranges of the packed matrix."""
    w_ih, w_hh, b_ih, b_hh = _params(3, i, h, dev, 11)
    gen = torch.Generator().manual_seed(m)
    x0 = torch.randn(m, i, generator=gen).to(dev)
    h0 = (torch.randn(m, h, generator=gen) * 0.5).to(dev)
    wgt = torch.randn(m, h, generator=gen).to(dev) * 1e-3          # gradients of a mean-like loss

    def run(packed):
        old, ops.GRU_PACKED_BACKWARD = ops.GRU_PACKED_BACKWARD, packed
        try:
            leaves = [t.clone().requires_grad_() for t in (x0, w_ih, w_hh, b_ih, b_hh)]
            out = ops.gru_cell(leaves[0], *leaves[1:], h0.clone(), None, mem_write=False)
            (out * wgt).sum().backward()
            return [t.grad for t in leaves]
        finally:
            ops.GRU_PACKED_BACKWARD = old

    cell = torch.nn.GRUCell(i, h).double()
    with torch.no_grad():
        for p, t in zip((cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh), (w_ih, w_hh, b_ih, b_hh)):
            p.copy_(t.double().cpu())
    xr = x0.double().cpu().requires_grad_()
    (cell(xr, h0.double().cpu()) * wgt.double().cpu()).sum().backward()
    want = [xr.grad, cell.weight_ih.grad, cell.weight_hh.grad, cell.bias_ih.grad, cell.bias_hh.grad]
    for packed in (True, False):
        for got, ref, name in zip(run(packed), want, ("x", "w_ih", "w_hh", "b_ih", "b_hh")):
            err = float((got.double().cpu() - ref).abs().max()) / float(ref.abs().max())
            assert err < 1e-5, (packed, name, err)


@pytest.mark.parametrize("m", [6000, 33, 4096 + 17, 20_000])
def test_tracked_whole_table_gru_step(dev, m):
    """The whole-table GRU kernel as a tracked step (gate values kept per state element, the old
    state read from the matrix it came from) against torch.nn.GRUCell in float64: new state and
    the gradients of x, both weights and both biases within 1e-5 of each tensor's largest entry;
    the state matrix it read is left as it was.  At 20000 rows both weight gradients come from
    one launch (spmi_gemm_tn_pair_sums)."""
    i = h = 128
    w_ih, w_hh, b_ih, b_hh = _params(3, i, h, dev, 21)
    gen = torch.Generator().manual_seed(m)
    x0 = torch.randn(m, i, generator=gen).to(dev)
    h0 = (torch.randn(m, h, generator=gen) * 0.5).to(dev)
    wgt = torch.randn(m, h, generator=gen).to(dev) * 1e-3
    leaves = [t.clone().requires_grad_() for t in (x0, w_ih, w_hh, b_ih, b_hh)]
    h_in = h0.clone()
    assert ops.gru_rows_tracked_ok(leaves[0], leaves[2])
    out = ops.gru_rows_tracked(leaves[0], *leaves[1:], h_in)
    assert out.data_ptr() != h_in.data_ptr() and torch.equal(h_in, h0)
    (out * wgt).sum().backward()

    cell = torch.nn.GRUCell(i, h).double()
    with torch.no_grad():
        for p, t in zip((cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh), (w_ih, w_hh, b_ih, b_hh)):
            p.copy_(t.double().cpu())
    xr = x0.double().cpu().requires_grad_()
    ref = cell(xr, h0.double().cpu())
    (ref * wgt.double().cpu()).sum().backward()
    err = float((out.detach().double().cpu() - ref.detach()).abs().max())
    assert err < 1e-5, err
    want = [xr.grad, cell.weight_ih.grad, cell.weight_hh.grad, cell.bias_ih.grad, cell.bias_hh.grad]
    for got, r, name in zip([t.grad for t in leaves], want, ("x", "w_ih", "w_hh", "b_ih", "b_hh")):
        e = float((got.double().cpu() - r).abs().max()) / float(r.abs().max())
        assert e < 1e-5, (name, e)
    # the same numbers as the inference kernel (one loop nest, one epilogue)
    with torch.no_grad():
        plain = ops.gru_rows(x0, w_ih, w_hh, b_ih, b_hh, h0, torch.empty_like(h0))
    assert torch.equal(plain, out.detach())


def test_tracked_whole_table_steps_through_the_decoder(dev):
    """Two tracked steps of the GRU decoder over every node (the train loop's shape): the memory
    adopts each step's output without a copy, a loss summed over both steps goes backward through
    both graphs (nothing a graph holds is overwritten by the second step), and the table read
    afterwards equals the second step's state.  Against the same decoder on the row-tile path."""
    from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import GRU
    from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP

    n, hd = 5000, 128

    def run(rows_kernel):
        old, ops.USE_GRU_ROWS = ops.USE_GRU_ROWS, rows_kernel
        try:
            torch.manual_seed(5)
            dec = GRU(hd, hd, n, 2, MLP(hd, [16], dropout=0.0)).to(dev)
            dec.train()
            torch.manual_seed(6)
            dec.memory_h.reset_state()
            dec.memory_h = dec.memory_h.to(dev)
            xs = [torch.randn(n, hd, device=dev, generator=torch.Generator(dev).manual_seed(40 + t)) for t in range(2)]
            loss = 0
            for x in xs:
                out = dec((x, list(range(n))))
                loss = loss + out.pow(2).mean()
            adopted = dec.memory_h._layers is not None
            loss.backward()
            grads = {k: p.grad.clone() for k, p in dec.named_parameters()}
            return float(loss.detach()), grads, dec.memory_h.memory.clone(), adopted
        finally:
            ops.USE_GRU_ROWS = old

    loss_a, grads_a, mem_a, adopted_a = run(True)
    loss_b, grads_b, mem_b, adopted_b = run(False)
    assert adopted_a and not adopted_b
    assert abs(loss_a - loss_b) <= 1e-5 * abs(loss_b)
    assert float((mem_a - mem_b).abs().max()) < 1e-5
    # against the model's largest gradient: the bias in front of a train-mode BatchNorm has a
    # gradient of exactly zero in exact arithmetic and rounding noise here
    scale = max(float(g.abs().max()) for g in grads_b.values())
    for k in grads_b:
        assert float((grads_a[k] - grads_b[k]).abs().max()) / scale < 1e-5, k



@pytest.mark.parametrize("m", [6000, 33, 4096 + 17, 20_000])
def test_tracked_whole_table_lstm_step(dev, m):
    """The whole-table LSTM kernel as a tracked step (six planes kept per state element, the old
    hidden state read from the matrix it came from) against torch.nn.LSTMCell in float64: both new
    states and the gradients of x, both weights and both biases within 1e-5 of each tensor's
    largest entry; the state matrices it read are left as they were.  At 20000 rows both weight
    gradients and the bias gradient come from one launch (spmi_gemm_tn_pair_sums)."""
    i = h = 128
    w_ih, w_hh, b_ih, b_hh = _params(4, i, h, dev, 22)
    gen = torch.Generator().manual_seed(m + 1)
    x0 = torch.randn(m, i, generator=gen).to(dev)
    h0 = (torch.randn(m, h, generator=gen) * 0.5).to(dev)
    c0 = torch.randn(m, h, generator=gen).to(dev)
    wgt = torch.randn(m, h, generator=gen).to(dev) * 1e-3
    leaves = [t.clone().requires_grad_() for t in (x0, w_ih, w_hh, b_ih, b_hh)]
    h_in, c_in = h0.clone(), c0.clone()
    assert ops.lstm_rows_tracked_ok(leaves[0], leaves[2])
    out, c_new = ops.lstm_rows_tracked(leaves[0], *leaves[1:], h_in, c_in)
    assert out.data_ptr() != h_in.data_ptr() and c_new.data_ptr() != c_in.data_ptr()
    assert torch.equal(h_in, h0) and torch.equal(c_in, c0) and not c_new.requires_grad
    (out * wgt).sum().backward()

    cell = torch.nn.LSTMCell(i, h).double()
    with torch.no_grad():
        for p, t in zip((cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh), (w_ih, w_hh, b_ih, b_hh)):
            p.copy_(t.double().cpu())
    xr = x0.double().cpu().requires_grad_()
    ref_h, ref_c = cell(xr, (h0.double().cpu(), c0.double().cpu()))
    (ref_h * wgt.double().cpu()).sum().backward()
    assert float((out.detach().double().cpu() - ref_h.detach()).abs().max()) < 1e-5
    assert float((c_new.double().cpu() - ref_c.detach()).abs().max()) < 1e-5 * float(ref_c.detach().abs().max())
    want = [xr.grad, cell.weight_ih.grad, cell.weight_hh.grad, cell.bias_ih.grad, cell.bias_hh.grad]
    for got, r, name in zip([t.grad for t in leaves], want, ("x", "w_ih", "w_hh", "b_ih", "b_hh")):
        e = float((got.double().cpu() - r).abs().max()) / float(r.abs().max())
        assert e < 1e-5, (name, e)
    # the same numbers as the inference kernel (one loop nest, one epilogue)
    with torch.no_grad():
        ph, pc = ops.lstm_rows(x0, w_ih, w_hh, b_ih, b_hh, h0, c0, torch.empty_like(h0), torch.empty_like(c0))
    assert torch.equal(ph, out.detach()) and torch.equal(pc, c_new)


@pytest.mark.parametrize("m", [6000, 33, 4096 + 17, 20_000])
def test_tracked_whole_table_rnn_step(dev, m):
    """The whole-table tanh-RNN kernel as a tracked step against torch.nn.RNNCell in float64: new
    state and the gradients of x, both weights and both biases within 1e-5 of each tensor's
    largest entry; the state matrix it read is left as it was."""
    i = h = 128
    w_ih, w_hh, b_ih, b_hh = _params(1, i, h, dev, 23)
    gen = torch.Generator().manual_seed(m + 2)
    x0 = torch.randn(m, i, generator=gen).to(dev)
    h0 = (torch.randn(m, h, generator=gen) * 0.5).to(dev)
    wgt = torch.randn(m, h, generator=gen).to(dev) * 1e-3
    leaves = [t.clone().requires_grad_() for t in (x0, w_ih, w_hh, b_ih, b_hh)]
    h_in = h0.clone()
    assert ops.rnn_rows_tracked_ok(leaves[0], leaves[2])
    out = ops.rnn_rows_tracked(leaves[0], *leaves[1:], h_in)
    assert out.data_ptr() != h_in.data_ptr() and torch.equal(h_in, h0)
    (out * wgt).sum().backward()

    cell = torch.nn.RNNCell(i, h).double()
    with torch.no_grad():
        for p, t in zip((cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh), (w_ih, w_hh, b_ih, b_hh)):
            p.copy_(t.double().cpu())
    xr = x0.double().cpu().requires_grad_()
    ref = cell(xr, h0.double().cpu())
    (ref * wgt.double().cpu()).sum().backward()
    assert float((out.detach().double().cpu() - ref.detach()).abs().max()) < 1e-5
    want = [xr.grad, cell.weight_ih.grad, cell.weight_hh.grad, cell.bias_ih.grad, cell.bias_hh.grad]
    for got, r, name in zip([t.grad for t in leaves], want, ("x", "w_ih", "w_hh", "b_ih", "b_hh")):
        e = float((got.double().cpu() - r).abs().max()) / float(r.abs().max())
        assert e < 1e-5, (name, e)


@pytest.mark.parametrize("kind", ["LSTM", "RNN"])
def test_tracked_whole_table_lstm_and_rnn_through_their_decoders(dev, kind):
    """Two tracked steps of the LSTM / RNN decoder over every node: the memories adopt each
    step's outputs without a copy, a loss summed over both steps goes backward through both
    graphs, and the tables read afterwards equal the second step's states.  Against the same
    decoder on the row-tile path."""
    from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder import implicitTimeModels as itm
    from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP

    n, hd = 5000, 128

    def run(rows_kernel):
        old, ops.USE_GRU_ROWS = ops.USE_GRU_ROWS, rows_kernel
        try:
            torch.manual_seed(7)
            dec = getattr(itm, kind)(hd, hd, n, 2, MLP(hd, [16], dropout=0.0)).to(dev)
            dec.train()
            torch.manual_seed(8)
            for name in dec._memories():
                getattr(dec, name).reset_state()
                setattr(dec, name, getattr(dec, name).to(dev))
            xs = [torch.randn(n, hd, device=dev, generator=torch.Generator(dev).manual_seed(60 + t)) for t in range(2)]
            loss = 0
            for x in xs:
                out = dec((x, list(range(n))))
                loss = loss + out.pow(2).mean()
            adopted = all(getattr(dec, name)._layers is not None for name in dec._memories())
            loss.backward()
            grads = {k: p.grad.clone() for k, p in dec.named_parameters()}
            return float(loss.detach()), grads, [getattr(dec, name).memory.clone() for name in dec._memories()], adopted
        finally:
            ops.USE_GRU_ROWS = old

    loss_a, grads_a, mems_a, adopted_a = run(True)
    loss_b, grads_b, mems_b, adopted_b = run(False)
    assert adopted_a and not adopted_b
    assert abs(loss_a - loss_b) <= 1e-5 * abs(loss_b)
    for ma, mb in zip(mems_a, mems_b):
        assert float((ma - mb).abs().max()) < 1e-5 * max(1.0, float(mb.abs().max()))
    scale = max(float(g.abs().max()) for g in grads_b.values())
    for k in grads_b:
        assert float((grads_a[k] - grads_b[k]).abs().max()) / scale < 1e-5, k

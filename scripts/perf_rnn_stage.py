"""The benchmark's recurrent stage (model.rnn of bench.Path) timed alone with events: with a
random input, and with the output of the two GraphConv layers on a generated snapshot."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spiral_amd import graph as G  # noqa: E402

dev = torch.device("cuda:0")
n, e, f = 1_000_000, 10_000_000, 128
torch.manual_seed(11)
model = bench.Path(f, 128, 128, n, dev)
ids = torch.arange(n, device=dev)
ids._spmi_identity = True


def timed(fn, iters=12):
    evs = []
    for i in range(iters + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs[2:])
    return ts[len(ts) // 2]


with torch.no_grad():
    h_rand = torch.randn(n, 128, device=dev)
    print("model.rnn on randn input: %.3f ms" % timed(lambda: model.rnn((h_rand, ids))))
    perm = torch.randperm(n, device=dev)
    src, dst = bench.gen_snapshot(n, e, 0, 3.0, 1.5, perm, dev)
    g = G.SnapshotGraph(src, dst, n)
    x = torch.randn(n, f, device=dev)
    h = model.conv2(g, model.conv1(g, x))
    print("conv output: mean |h| %.4g, max |h| %.4g, fraction |h| < 2^-8: %.3f"
          % (h.abs().mean(), h.abs().max(), (h.abs() < 2.0 ** -8).float().mean()))
    print("model.rnn on conv output: %.3f ms" % timed(lambda: model.rnn((h, ids))))
    mem = model.rnn.memory if hasattr(model.rnn, "memory") else None
    hs = h * 0.01
    print("model.rnn on conv output * 0.01: %.3f ms" % timed(lambda: model.rnn((hs, ids))))
    hb = h * 100
    print("model.rnn on conv output * 100: %.3f ms" % timed(lambda: model.rnn((hb, ids))))

# bisect: the same layer step through ops.gru_cell with the model's own tensors
from spiral_amd import ops  # noqa: E402

with torch.no_grad():
    rnn = model.rnn
    params = rnn._layer_params(0)
    mem0 = rnn.memory_h.memory[0]
    print("training_mode", rnn.training_mode, "mem", tuple(rnn.memory_h.memory.shape), rnn.memory_h.memory.dtype,
          "contiguous", mem0.is_contiguous(), "ptr % 512", mem0.data_ptr() % 512,
          "param requires_grad", params[0].requires_grad, "w_ih", tuple(params[0].shape))
    print("ops.gru_cell, model tensors, mem_write as the model: %.3f ms"
          % timed(lambda: ops.gru_cell(h_rand, *params, mem0, None, mem_write=rnn.training_mode)))
    print("ops.gru_cell, model tensors, mem_write=True: %.3f ms"
          % timed(lambda: ops.gru_cell(h_rand, *params, mem0, None, mem_write=True)))
    det = [p.detach().clone() for p in params]
    mem1 = torch.randn_like(mem0)
    print("ops.gru_cell, cloned params and fresh memory: %.3f ms"
          % timed(lambda: ops.gru_cell(h_rand, *det, mem1, None, mem_write=True)))
    big = [p.detach().clone() * 0 + torch.randn_like(p) * 0.05 for p in params]
    print("ops.gru_cell, randn*0.05 params and fresh memory: %.3f ms"
          % timed(lambda: ops.gru_cell(h_rand, *big, mem1, None, mem_write=True)))

"""Forward + backward of a two-layer recurrent decoder step over every node under autograd, on
the whole-table kernels (tracked steps) and on the row-tile path.
usage: perf_tracked_cells.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder import implicitTimeModels as itm  # noqa: E402
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
hd = 128
ids = None  # every node


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for kind in ("GRU", "LSTM", "RNN"):
    line = []
    for whole in (True, False):
        ops.USE_GRU_ROWS = whole
        torch.manual_seed(0)
        dec = getattr(itm, kind)(hd, hd, rows, 2, MLP(hd, [16], dropout=0.0)).to(dev)
        dec.train()
        for name in dec._memories():
            setattr(dec, name, getattr(dec, name).to(dev))
        x = torch.randn(rows, hd, device=dev, requires_grad=True)

        def step():
            for p in dec.parameters():
                p.grad = None
            dec((x, ids)).pow(2).mean().backward()

        line.append(timed(step))
        del dec
    ops.USE_GRU_ROWS = True
    print(f"{kind}: forward + backward of a 2-layer step over {rows} rows: whole-table kernels {line[0]:.2f} ms, "
          f"row-tile path {line[1]:.2f} ms")

"""Cost of the row-magnitude pass of the recurrent kernel: the same GRU/LSTM/RNN step with and
without the caller's promise that |x| < 2^15 (1M nodes, I = H = 128, inference)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402

def main():
    dev = torch.device("cuda:0")
    m, i, h = 1_000_000, 128, 128
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(m, i, device=dev, generator=g)
    for cell, gates in (("gru", 3), ("lstm", 4), ("rnn", 1)):
        w_ih = torch.randn(gates * h, i, device=dev, generator=g) * 0.1
        w_hh = torch.randn(gates * h, h, device=dev, generator=g) * 0.1
        b_ih = torch.randn(gates * h, device=dev, generator=g) * 0.1
        b_hh = torch.randn(gates * h, device=dev, generator=g) * 0.1
        mem = torch.randn(m, h, device=dev, generator=g)
        cmem = torch.randn(m, h, device=dev, generator=g)
        for rep in range(2):
            for bounded in (False, True):
                def step():
                    with torch.no_grad():
                        if cell == "gru":
                            ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mem, mem_write=False, x_bounded=bounded)
                        elif cell == "lstm":
                            ops.lstm_cell(x, w_ih, w_hh, b_ih, b_hh, mem, cmem, mem_write=False, x_bounded=bounded)
                        else:
                            ops.rnn_cell(x, w_ih, w_hh, b_ih, b_hh, mem, mem_write=False, x_bounded=bounded)
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(30):
                    step()
                b.record()
                torch.cuda.synchronize()
                print(f"{cell} x_bounded={bounded}: {a.elapsed_time(b) / 30:.3f} ms", flush=True)

if __name__ == "__main__":
    main()

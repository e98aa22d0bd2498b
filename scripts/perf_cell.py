"""Timing of one recurrent layer step: fused MFMA kernel vs library GEMMs + gate kernel."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import ops  # noqa: E402


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--i", type=int, default=128)
    ap.add_argument("--h", type=int, default=128)
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--gemm", default="f16x3", choices=["f16x3", "bf16x3", "fp32"])
    a = ap.parse_args()
    ops.CELL_GEMM = a.gemm
    dev = torch.device("cuda:0")
    x = torch.randn(a.n, a.i, device=dev)
    for kind, g in (("gru", 3), ("lstm", 4), ("rnn", 1)):
        if a.only and kind != a.only:
            continue
        w_ih = torch.randn(g * a.h, a.i, device=dev) * 0.05
        w_hh = torch.randn(g * a.h, a.h, device=dev) * 0.05
        b_ih = torch.randn(g * a.h, device=dev) * 0.05
        b_hh = torch.randn(g * a.h, device=dev) * 0.05
        mem_h = torch.randn(a.n, a.h, device=dev)
        mem_c = torch.randn(a.n, a.h, device=dev)
        flops = 2 * a.n * (a.i + a.h) * g * a.h
        for fused in ((True,) if a.only else (False, True)):
            ops.USE_FUSED_CELLS = fused

            def run():
                with torch.no_grad():
                    if kind == "gru":
                        ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mem_h)
                    elif kind == "lstm":
                        ops.lstm_cell(x, w_ih, w_hh, b_ih, b_hh, mem_h, mem_c)
                    else:
                        ops.rnn_cell(x, w_ih, w_hh, b_ih, b_hh, mem_h)

            ms = timed(run, a.iters)
            print(f"{kind} fused={fused}: {ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()

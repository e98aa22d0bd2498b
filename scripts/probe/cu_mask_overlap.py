"""Do the aggregation (HBM-bound) and the whole-table GRU (matrix/issue-bound, one workgroup per
CU) run side by side when each gets a share of the CUs?  Streams with a CU mask
(hipExtStreamCreateWithCUMask); the mask takes the same CUs of every XCD whichever way the
bits enumerate them (bit i -> XCD i % 8 or XCD i / 32)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from spiral_amd import graph as G, ops  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(keep):
    """keep(c) for c = 0..31, the CU's index inside its XCD under either enumeration."""
    bits = 0
    for i in range(256):
        if keep((i // 8) % 32) :
            bits |= 1 << i
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * w)) & 0xFFFFFFFF for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value), bin(bits).count("1")


dev = torch.device("cuda:0")
n, e, h, g = 1_000_000, 10_000_000, 128, 3
perm = torch.randperm(n, device=dev)
src, dst = bench.gen_snapshot(n, e, 0, 3.0, 1.5, perm, dev)
x = torch.randn(n, h, device=dev)
w1, w2 = torch.randn(g * h, h, device=dev) * 0.05, torch.randn(g * h, h, device=dev) * 0.05
b = torch.randn(g * h, device=dev) * 0.05
mem = torch.randn(2, n, h, device=dev)
csr = G.build_csr(dst, src, n, n, G.HEAVY_THRESH, G.HEAVY_CHUNK)
csr_t = G.build_csr(src, dst, n, n, G.HEAVY_THRESH, G.HEAVY_CHUNK)
ss, ds = csr_t.norm(0), csr.norm(0)
bias = torch.zeros(h, device=dev)
out = torch.empty(n, h, device=dev)


def agg():
    G.aggregate(csr, x, ss, None, ds, bias, "leaky_relu", 0.01, out=out)


def cell():
    ops.gru_rows(x, w1, w2, b, b, mem[0], mem[1], x_bounded=True)


def wall(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


with torch.no_grad():
    print(f"whole chip: aggregation {wall(agg):.3f} ms, GRU layer {wall(cell):.3f} ms", flush=True)
    for k in (8, 12, 16, 20, 24):  # CUs of every XCD's 32 that the aggregation gets
        sa, na = masked_stream(lambda c: c < k)
        sc, nc = masked_stream(lambda c: c >= k)

        def on(stream, fn):
            def run():
                with torch.cuda.stream(stream):
                    fn()
            return run

        def both():
            with torch.cuda.stream(sa):
                agg()
            with torch.cuda.stream(sc):
                cell()

        ta, tc_, tb = wall(on(sa, agg)), wall(on(sc, cell)), wall(both)
        print(f"aggregation on {na} CUs {ta:.3f} ms, GRU on {nc} CUs {tc_:.3f} ms, side by side {tb:.3f} ms", flush=True)

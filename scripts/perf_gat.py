"""Times GATConv (forward, and forward + backward) on the 1M-node / 10M-edge power-law snapshot."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import graph as G
from spiral_amd.core.types import GATConv


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def main():
    n, e = 1_000_000, 10_000_000
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(3)
    perm = torch.randperm(n, generator=gen, device=dev)
    dst = perm[(n * torch.rand(e, device=dev, generator=gen) ** 3).long().clamp_(max=n - 1)]
    src = perm[(n * torch.rand(e, device=dev, generator=gen) ** 1.5).long().clamp_(max=n - 1)]
    g = G.SnapshotGraph(src, dst, n).add_self_loop()
    g.csr(want_eid=True), g.csr_t(want_eid=True)
    for heads, d in ((1, 128), (4, 32), (2, 64)):
        conv = GATConv(128, d, heads, activation=F.leaky_relu).to(dev)
        x = torch.randn(n, 128, device=dev, generator=gen)
        with torch.no_grad():
            t_fwd = timed(lambda: conv(g, x))
        xg = x.clone().requires_grad_()

        def train():
            conv.zero_grad(set_to_none=True)
            xg.grad = None
            conv(g, xg).sum().backward()

        t_train = timed(train, reps=3)
        print(f"GATConv 128 -> {heads}x{d}: forward {t_fwd:.2f} ms, forward+backward {t_train:.2f} ms", flush=True)


if __name__ == "__main__":
    main()

"""Quick timing of the fused aggregation and the CSR build on one snapshot."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import graph as G  # noqa: E402


def powerlaw(n, e, t, k_dst, k_src, dev):
    rng = np.random.Generator(np.random.PCG64(1234 + t))
    perm = np.random.Generator(np.random.PCG64(99)).permutation(n)
    dst = perm[np.minimum((n * rng.random(e) ** k_dst).astype(np.int64), n - 1)]
    src = perm[np.minimum((n * rng.random(e) ** k_src).astype(np.int64), n - 1)]
    return torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--e", type=int, default=10_000_000)
    ap.add_argument("--f", type=int, default=128)
    ap.add_argument("--k-dst", type=float, default=3.0)
    ap.add_argument("--k-src", type=float, default=1.5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--thresh", type=int, default=G.HEAVY_THRESH)
    ap.add_argument("--chunk", type=int, default=G.HEAVY_CHUNK)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t0 = time.time()
    src, dst = powerlaw(a.n, a.e, 0, a.k_dst, a.k_src, dev)
    print(f"gen {time.time()-t0:.2f}s", flush=True)
    med, best = timed(lambda: G.build_csr(dst, src, a.n, a.n, a.thresh, a.chunk), 5)
    print(f"build_csr(+plan, sync): median {med:.3f} ms best {best:.3f} ms")
    csr = G.build_csr(dst, src, a.n, a.n, a.thresh, a.chunk)
    csr_t = G.build_csr(src, dst, a.n, a.n, a.thresh, a.chunk)
    print(f"n_heavy {csr.n_heavy} n_chunks {csr.n_chunks} max_deg {int(csr.degrees().max())}")
    x = torch.randn(a.n, a.f, device=dev)
    sb = 4
    if a.dtype == 'bf16':
        x = x.to(torch.bfloat16)
        sb = 2
    ss, ds = csr_t.norm(0), csr.norm(0)
    bias = torch.zeros(a.f, device=dev)
    ew = ss.index_select(0, csr.indices)
    med, best = timed(lambda: ss.index_select(0, csr.indices), 5)
    print(f'edge weight gather: {med:.3f} ms')
    out = torch.empty(a.n, a.f, device=dev, dtype=x.dtype)
    bytes_alg = a.e * (a.f * sb + 4) + (a.n + 1) * 4 + 2 * a.n * 4 + a.n * a.f * sb + a.f * 4
    for name, fn in (
        ("spmm plain", lambda: G.aggregate(csr, x, out=out)),
        ("spmm fused norm+bias+act", lambda: G.aggregate(csr, x, ss, None, ds, bias, "leaky_relu", 0.01, out=out)),
        ("spmm fused, per-edge weight stream", lambda: G.aggregate(csr, x, None, ew, ds, bias, "leaky_relu", 0.01, out=out)),
    ):
        med, best = timed(fn, a.iters)
        print(f"{name}: median {med:.3f} ms best {best:.3f} ms  {a.e/med/1e6:.2f} Gedges/s  "
              f"{bytes_alg/med/1e9:.2f} TB/s algorithmic ({bytes_alg/med/1e9/8*100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()

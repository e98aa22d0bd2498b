"""Times the per-snapshot CSR cut (table-wide grouping + row prefixes) against a fresh sort."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import graph as G


def timed(fn, reps=10):
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
    g = torch.Generator(device=dev).manual_seed(3)
    dst = (n * torch.rand(e, device=dev, generator=g) ** 3).long().clamp_(max=n - 1)
    src = (n * torch.rand(e, device=dev, generator=g) ** 1.5).long().clamp_(max=n - 1)
    cum = G.CumulativeCSR(dst, src, n, n)
    for frac in (1.0, 0.5, 0.1):
        c = int(e * frac)
        t_sort = timed(lambda: G.build_csr(dst[:c], src[:c], n, n))
        t_cut = timed(lambda: cum.prefix(c, n, n))
        t_cut_eid = timed(lambda: cum.prefix(c, n, n, want_eid=True))
        t_deg = timed(lambda: cum.prefix_degree(c, n))
        a, b = G.build_csr(dst[:c], src[:c], n, n), cum.prefix(c, n, n)
        same = torch.equal(a.indptr, b.indptr) and torch.equal(a.indices, b.indices)
        print(f"prefix {frac:4.0%}: fresh sort {t_sort:.3f} ms | cut {t_cut:.3f} ms | cut+eid "
              f"{t_cut_eid:.3f} ms | degrees only {t_deg:.3f} ms | identical={same}", flush=True)


if __name__ == "__main__":
    main()

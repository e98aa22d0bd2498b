"""Times the paired weight-gradient launch on the operands a real train step hands it (captured
from bench.py --mode train) beside random operands of the same shape, and prints how far the
16-row blocks' magnitudes spread.  usage: capture_pair.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spiral_amd import ops  # noqa: E402

captured = []
real = ops.gemm_tn_pair


def spy(a0, b0, a1, b1, sums=False):
    if len(captured) < 2:
        # a0 and a1 are views of the step's packed gate gradients: kept as they are (strides
        # included), the bench holds no reference that would overwrite them before the timing
        captured.append((a0, b0.clone(), a1, b1.clone(), sums))
    return real(a0, b0, a1, b1, sums=sums)


ops.gemm_tn_pair = spy
import bench  # noqa: E402

sys.argv = ["bench.py", "--mode", "train", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
bench.main()
ops.gemm_tn_pair = real
print("captured", len(captured), file=sys.stderr)


def timed(fn, n=10):
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


for idx, (a0, b0, a1, b1, sums) in enumerate(captured):
    print(f"call {idx}: a0 {tuple(a0.shape)} stride {a0.stride()} b0 {tuple(b0.shape)} stride {b0.stride()} "
          f"a1 {tuple(a1.shape)} stride {a1.stride()} b1 {tuple(b1.shape)} sums {sums}")
    t_real = timed(lambda: real(a0, b0, a1, b1, sums=sums))
    for name, a, b in (("a0*b0", a0, b0), ("a1*b1", a1, b1)):
        k = a.shape[0] // 16 * 16
        ea = a[:k].abs().reshape(k // 16, -1).amax(1).clamp_min(1e-45).log2().floor()
        eb = b[:k].abs().reshape(k // 16, -1).amax(1).clamp_min(1e-45).log2().floor()
        t = ea + eb
        print(f"  {name}: exponent sums of 16-row blocks min {t.min().item():.0f} median {t.median().item():.0f} "
              f"max {t.max().item():.0f}; |a| max {a.abs().max().item():.3e} |b| max {b.abs().max().item():.3e}")
    g = torch.randn(a0.shape[0], 512, device=a0.device) * 1e-3
    x = torch.randn_like(b0)
    h = torch.randn_like(b1)
    t_rand = timed(lambda: real(g[:, :384], x, g[:, 128:], h, sums=sums))
    print(f"  one launch: captured operands {t_real:.3f} ms, random operands {t_rand:.3f} ms")

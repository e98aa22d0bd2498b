"""Rehearsal of bench.py's fallback: the asynchronous gather is made to raise, every rank must
switch to the plain exchange and still print a valid line.  Launch like bench.py (torchrun,
SPMI_BENCH_REHEARSE=1 on a one-GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spiral_amd import parallel as par  # noqa: E402


def _broken(*args, **kwargs):
    raise RuntimeError("injected failure of the asynchronous gather")


par.gather_rows_async = _broken
bench.main()

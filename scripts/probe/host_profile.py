"""cProfile of bench.py on a workload whose step is bound by the host (default: small): where
the Python between the kernels goes.  usage: host_profile.py [workload] [steps] [forward|train]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "small"
steps = sys.argv[2] if len(sys.argv) > 2 else "400"
mode = sys.argv[3] if len(sys.argv) > 3 else "forward"
sys.argv = ["bench.py", "--workload", workload, "--steps", steps, "--warmup", "4", "--no-cpu-baseline",
            "--no-train-probe", "--mode", mode]
prof = cProfile.Profile()
prof.enable()
bench.main()
prof.disable()
st = pstats.Stats(prof, stream=sys.stderr)
st.sort_stats("tottime").print_stats(45)

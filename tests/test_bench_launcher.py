"""`python bench.py --gpus N` must start its own N ranks when no launcher did.

The CPU cases run the launcher's plumbing alone (SPMI_BENCH_LAUNCH_CHECK=1: rendezvous over
gloo, one collective, rank 0's JSON line, exit status), which needs no GPU.  The GPU case runs
the small workload with two ranks sharing the one GPU of a test box (rehearsal mode: gloo with
host staging) and checks that the multi-rank line prices the exchange plan."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    proc = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    return proc, (json.loads(lines[-1]) if lines else None)


def test_self_launch_starts_the_ranks_and_returns_rank_zeros_line():
    proc, line = _run(["--gpus", "2"], {"SPMI_BENCH_LAUNCH_CHECK": "1"})
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert line == {"launch_check": True, "world": 2, "rank_sum": 3, "launched_by_bench": True}
    assert sum(l.startswith("{") for l in proc.stdout.splitlines()) == 1   # one JSON line on stdout


def test_self_launch_reports_a_failing_rank():
    proc, _ = _run(["--gpus=3"], {"SPMI_BENCH_LAUNCH_CHECK": "1", "SPMI_BENCH_LAUNCH_CHECK_FAIL_RANK": "2",
                                  "SPMI_BENCH_LAUNCH_GRACE_S": "2"})
    assert proc.returncode == 7
    assert "rank 2 exited with status 7" in proc.stderr


def test_launcher_environment_is_left_alone():
    """Under torch.distributed.run (RANK / WORLD_SIZE set) every process is a rank: no second launch."""
    env = {"SPMI_BENCH_LAUNCH_CHECK": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29733"}
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=dict(os.environ, **env),
                          capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][-1])
    assert line["world"] == 1 and line["launched_by_bench"] is False


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--exchange", "pipelined"], ["--exchange", "plain", "--scaling", "strong"],
                                   ["--mode", "train", "--stages", "2"]])
def test_two_ranks_through_the_self_launcher_price_the_plan(extra):
    proc, line = _run(["--gpus", "2", "--workload", "small", "--steps", "3", "--warmup", "2",
                       "--no-cpu-baseline"] + extra, {"SPMI_BENCH_REHEARSE": "1"}, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["value"] > 0
    st = line["stages"]
    for key in ("plan_ms", "csr_build_ms", "out_degree_exchange_ms", "recurrent_ms"):
        assert st[key] >= 0.0, key
    ex = line["exchange_stats"]
    assert len(ex["per_rank"]) == 2
    for r in ex["per_rank"]:
        assert 0 < r["rows_received"] <= ex["all_gather_rows_per_rank"] * 1.01 + 1
        assert r["bytes_received"] == r["rows_received"] * ex["row_bytes"]
    assert 1.0 <= ex["nnz_imbalance"] < 1.5
    if "train" in extra:
        assert line["config"]["exchange"] == "batched"

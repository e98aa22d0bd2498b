"""The example pipeline (examples/citation_prediction.py: the reference's
examples/torch/citation_prediction.py with the package name swapped, on a synthetic citation
dataset) runs end to end for every encoder / cell pairing it offers."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                      "citation_prediction.py")


@pytest.mark.parametrize("gnn,rnn", [("gcn", "gru"), ("gat", "lstm"), ("sgcn", "rnn"), ("sage", "gru")])
def test_example_runs(monkeypatch, capsys, gnn, rnn):
    monkeypatch.setattr(sys, "argv", [SCRIPT, "--epochs", "1", "--nodes", "3000", "--edges", "20000",
                                      "--gnn", gnn, "--rnn", rnn])
    runpy.run_path(SCRIPT, run_name="__main__")
    out = capsys.readouterr().out
    assert "epoch 0: train loss" in out and "nan" not in out.lower()

"""The import lines of the reference's pipeline (examples/torch/citation_prediction.py:7-14) with
the package name swapped resolve to the classes on the path."""
import pytest


def test_reference_pipeline_imports():
    from spiral_amd.core.backends import PYTORCH, TENSORFLOW
    from spiral_amd.core.config import get_backend, set_backend
    from spiral_amd.core.errors import BackendNotSupportedError

    set_backend(PYTORCH)
    assert get_backend() == PYTORCH
    with pytest.raises(BackendNotSupportedError):
        set_backend(TENSORFLOW)
    from spiral_amd.automl.model_assembler import assembler
    from spiral_amd.dtdg.dataloader import ogb_dataset, supported_ogb_datasets  # :10, used at :32-33
    from spiral_amd.dtdg.models.decoder.sequentialDecoders import GRU, LSTM, RNN
    from spiral_amd.dtdg.models.decoder.simpleDecoders import MLP
    from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder import implicitTimeModels as itm
    from spiral_amd.dtdg.models.encoder.implicitTimeEncoder.staticGraphEncoder import GAT, GCN, SGCN, GraphSage

    assert (GRU, LSTM, RNN) == (itm.GRU, itm.LSTM, itm.RNN)
    assert all(callable(c) for c in (assembler, MLP, GCN, GAT, SGCN, GraphSage, ogb_dataset))
    assert "ogbn-arxiv" in supported_ogb_datasets()


def test_backend_selection_as_the_reference_tests_it(monkeypatch):
    """tests/core/test_config.py:25-35 of the reference: an invalid name falls back to PyTorch,
    importing the backend_config module selects it; the choice is mirrored in DGLBACKEND."""
    import importlib
    import os

    from spiral_amd.core.backends import PYTORCH, TENSORFLOW, supported_backends
    from spiral_amd.core.config import get_backend, set_backend

    assert supported_backends() == [PYTORCH]  # the reference lists TensorFlow too; here it raises
    monkeypatch.delenv("DGLBACKEND", raising=False)
    set_backend("bbb")
    assert os.environ["DGLBACKEND"] == PYTORCH and get_backend() == PYTORCH
    monkeypatch.delenv("DGLBACKEND", raising=False)
    import spiral_amd.core.backend_config.torch as by_code
    importlib.reload(by_code)
    assert os.environ["DGLBACKEND"] == PYTORCH


def test_small_helpers_under_the_reference_names(capsys):
    import numpy as np
    import torch

    from spiral_amd.core import errors
    from spiral_amd.core.torch_functions import _to_tensor, nt_layers_list, reshape_tensor
    from spiral_amd.core.utils import _get_backend, printProgressBar

    assert _get_backend() == "pytorch"
    assert torch.equal(reshape_tensor(_to_tensor(np.arange(6)), [2, 3]), torch.arange(6).reshape(2, 3))
    assert isinstance(nt_layers_list(), torch.nn.ModuleList)
    assert issubclass(errors.ValueError, Exception) and errors.ValueError is not ValueError
    printProgressBar(1, 4, prefix="p", suffix="s", length=8)
    printProgressBar(4, 4, prefix="p", suffix="s", length=8)
    out = capsys.readouterr().out
    assert "p |██------| 25.0% s" in out and "p |████████| 100.0% s" in out and out.endswith("\n")

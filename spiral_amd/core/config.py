"""`set_backend` as called at the top of the reference's pipelines (spiro/core/config.py:47-65,
e.g. examples/torch/citation_prediction.py:7-9).  The reference falls back to PyTorch with a
warning when the name is unknown; here PyTorch is the only backend, so TensorFlow raises."""
from spiral_amd.core.backends import PYTORCH, TENSORFLOW, supported_backends
from spiral_amd.core.errors import BackendNotSupportedError

_BACKEND = PYTORCH


def set_backend(backend: str) -> None:
    if backend == TENSORFLOW:
        raise BackendNotSupportedError("Backend %s not supported." % backend)
    global _BACKEND
    _BACKEND = PYTORCH if backend not in supported_backends() else backend


def get_backend() -> str:
    return _BACKEND

"""`set_backend` as called at the top of the reference's pipelines (spiro/core/config.py:47-65,
e.g. examples/torch/citation_prediction.py:7-9).  As there, an unknown name falls back to
PyTorch with a warning and the choice is also left in the DGLBACKEND environment variable (a
pipeline that imports dgl afterwards reads it).  PyTorch is the only backend here, so
TensorFlow raises."""
import logging
import os

from spiral_amd.core.backends import PYTORCH, TENSORFLOW, supported_backends
from spiral_amd.core.errors import BackendNotSupportedError

_DGL_BACKEND = "DGLBACKEND"
_BACKEND = PYTORCH


def set_backend(backend: str) -> None:
    if backend == TENSORFLOW:
        raise BackendNotSupportedError("Backend %s not supported." % backend)
    name = backend
    if name not in supported_backends():
        logging.getLogger(__name__).warning(
            "backend %r is not one of %s: assuming %s", backend, supported_backends(), PYTORCH)
        name = PYTORCH
    os.environ[_DGL_BACKEND] = name
    global _BACKEND
    _BACKEND = name


def get_backend() -> str:
    return _BACKEND

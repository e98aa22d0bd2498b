"""Importing this module selects the PyTorch backend (reference:
spiro/core/backend_config/torch.py:24-27), the one backend this package has."""
from spiral_amd.core.backends import PYTORCH
from spiral_amd.core.config import set_backend

set_backend(PYTORCH)

"""Backend names as in spiro/core/backends.py:26-36.  Only the PyTorch backend exists here."""
from typing import List

TENSORFLOW = "tensorflow"
PYTORCH = "pytorch"


def supported_backends() -> List[str]:
    return [PYTORCH]

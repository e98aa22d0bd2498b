"""PyTorch versions of the backend-dependent helpers under the module name the reference keeps
them in (spiro/core/torch_functions.py:24-44); `core.commonF` re-exports them."""
from spiral_amd.core.commonF import nt_layers_list, reshape_tensor
from spiral_amd.core.commonF import to_tensor as _to_tensor

__all__ = ["_to_tensor", "nt_layers_list", "reshape_tensor"]

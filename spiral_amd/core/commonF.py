"""Tensor helpers named as in spiro/core/commonF.py:24-33 (torch backend only)."""
from typing import List

import torch
from numpy import ndarray


def to_tensor(arr: ndarray) -> torch.Tensor:
    return torch.from_numpy(arr)


def reshape_tensor(h: torch.Tensor, new_shape: List[int]) -> torch.Tensor:
    return torch.reshape(h, new_shape)


def nt_layers_list() -> torch.nn.ModuleList:
    return torch.nn.ModuleList()

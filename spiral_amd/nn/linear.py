"""torch.nn.Linear whose weight gradient, dW = dY^T X over millions of node rows, runs on
the split-K matrix-core kernel of libspmi (spmi_gemm_tn) instead of a library GEMM.
Same parameters, initialisation and state_dict as torch.nn.Linear."""
from __future__ import annotations

import torch
from torch import nn

from spiral_amd.ops import WeightCacheHooks, linear_rows


class TallLinear(WeightCacheHooks, nn.Linear):
    def forward(self, input: torch.Tensor) -> torch.Tensor:  # noqa: A002 (torch's argument name)
        if input.is_cuda and input.dim() == 2 and input.dtype == torch.float32:
            return linear_rows(input, self.weight, self.bias)
        return super().forward(input)

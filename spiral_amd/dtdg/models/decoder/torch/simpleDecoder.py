"""Output heads; API of spiro/dtdg/models/decoder/torch/simpleDecoder.py:24-79.
Widths are tens of units, so this stays plain torch (library GEMMs)."""
from abc import abstractmethod
from typing import List, Tuple

import torch
from torch import nn


class SimpleDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList()

    @abstractmethod
    def forward(self, in_sample: Tuple[torch.Tensor, List]) -> torch.Tensor:
        ...


class MLP(SimpleDecoder):
    """[Linear, BatchNorm1d, ReLU, Dropout] per entry of embed_dims, then Linear to
    output_dim.  The node ids of the input pair are not used (simpleDecoder.py:78-79)."""

    def __init__(self, input_dim: int, embed_dims: List[int], dropout: float = 0.5, output_dim: int = 1):
        super().__init__()
        width = input_dim
        for embed_dim in embed_dims:
            self.layers.extend([nn.Linear(width, embed_dim), nn.BatchNorm1d(embed_dim), nn.ReLU(),
                                nn.Dropout(p=dropout)])
            width = embed_dim
        self.layers.append(nn.Linear(width, output_dim))
        self.mlp = nn.Sequential(*self.layers)

    def forward(self, in_state):
        node_embs, _ids = in_state
        return self.mlp(node_embs)

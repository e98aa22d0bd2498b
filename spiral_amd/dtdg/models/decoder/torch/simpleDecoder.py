"""Output heads; API of spiro/dtdg/models/decoder/torch/simpleDecoder.py:24-79.
Widths are tens of units, so this stays plain torch (library GEMMs)."""
from abc import abstractmethod
from typing import List, Tuple

import torch
from torch import nn

from spiral_amd.nn.linear import TallLinear


class SimpleDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList()

    @abstractmethod
    def forward(self, in_sample: Tuple[torch.Tensor, List]) -> torch.Tensor:
        ...


_ACTIVATIONS = {"linear": None, None: None, "sigmoid": nn.Sigmoid, "relu": nn.ReLU, "tanh": nn.Tanh}


class MLP(SimpleDecoder):
    """[Linear, BatchNorm1d, ReLU, Dropout] per entry of embed_dims, then Linear to
    output_dim.

    Node task: ids of rank 1 are not used (simpleDecoder.py:78-79).  Edge task: ids of
    rank 2, shape (n_edges, 2), index rows of the embedding; the two rows of a pair are
    concatenated before the first layer, so input_dim is twice the embedding width.  The
    reference implements the edge task, and the `activation` of the last layer, only in
    its TensorFlow MLP (spiro/dtdg/models/decoder/tf/simpleDecoder.py:45-97)."""

    def __init__(self, input_dim: int, embed_dims: List[int], dropout: float = 0.5, output_dim: int = 1,
                 activation: str = "linear"):
        super().__init__()
        if activation not in _ACTIVATIONS:
            raise ValueError(f"unsupported activation {activation!r}")
        width = input_dim
        for embed_dim in embed_dims:
            self.layers.extend([TallLinear(width, embed_dim), nn.BatchNorm1d(embed_dim), nn.ReLU(),
                                nn.Dropout(p=dropout)])
            width = embed_dim
        self.layers.append(TallLinear(width, output_dim))
        if _ACTIVATIONS[activation] is not None:
            self.layers.append(_ACTIVATIONS[activation]())
        self.mlp = nn.Sequential(*self.layers)

    def forward(self, in_state):
        node_embs, ids = in_state
        if isinstance(ids, torch.Tensor) and ids.dim() == 2:
            pairs = ids.to(node_embs.device).reshape(-1)
            node_embs = node_embs.index_select(0, pairs).reshape(ids.shape[0], -1)
        elif isinstance(ids, torch.Tensor) and ids.dim() > 2:
            from spiral_amd.core.errors import DimensionError

            raise DimensionError("The index to predict in the input must be of rank 1 for node prediction "
                                 f"or rank 2 for edge prediction. But get an input index of rank {ids.dim()}")
        return self.mlp(node_embs)

"""The single import point of the layer types, as spiro/core/types.py:33-40 is for the
reference: swapping `from spiro.core.types import GraphConv` for this module swaps DGL's
GraphConv for the libspmi-backed one."""
from torch import Tensor as Tensor
from torch.nn import Dropout as Dropout
from torch.nn import Module as MLBaseModel
from torch.nn import Sequential as Sequential

from spiral_amd.core.commonF import nt_layers_list
from spiral_amd.nn.gatconv import GATConv
from spiral_amd.nn.graphconv import GraphConv
from spiral_amd.nn.sageconv import SAGEConv
from spiral_amd.nn.sgconv import SGConv

__all__ = ["Tensor", "Dropout", "MLBaseModel", "Sequential", "nt_layers_list", "GraphConv", "SGConv",
           "SAGEConv", "GATConv"]

from .evolve import EvolveGCNO, WeightGRU
from .gatconv import GATConv
from .graphconv import GraphConv
from .sageconv import SAGEConv
from .sgconv import SGConv

__all__ = ["GraphConv", "SGConv", "SAGEConv", "GATConv", "EvolveGCNO", "WeightGRU"]

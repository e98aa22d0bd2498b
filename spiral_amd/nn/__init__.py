from .graphconv import GraphConv

__all__ = ["GraphConv"]

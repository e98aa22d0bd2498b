"""Exception types raised on the snapshot path; names follow spiro/core/errors.py:18-31."""


class DimensionError(Exception):
    """An input table has fewer columns than the container needs."""


class ValueNotSortedError(Exception):
    """A table or the timestamp index is not sorted by time."""


class GraphError(Exception):
    """Raised where the reference surfaces dgl.DGLError (invalid norm, zero in-degree...)."""


DGLError = GraphError


class BackendNotSupportedError(Exception):
    """A backend other than PyTorch was asked for (spiro/core/errors.py; the TensorFlow
    backend of the reference is not part of this library)."""


class ValueError(Exception):  # noqa: A001  (the reference's name, spiro/core/errors.py:42)
    """The package's own value error; shadows the builtin inside this module only."""

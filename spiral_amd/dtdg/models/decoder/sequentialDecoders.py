"""Import surface of spiro/dtdg/models/decoder/sequentialDecoders.py (the backend switch of
the reference; the PyTorch decoders are the ones that exist here)."""
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import (  # noqa: F401
    GRU,
    LSTM,
    RNN,
)

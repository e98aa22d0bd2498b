"""Import surface of spiro/dtdg/models/decoder/simpleDecoders.py."""
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP as MLP  # noqa: F401

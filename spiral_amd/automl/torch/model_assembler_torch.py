"""Encoder + decoder glue (spiro/automl/torch/model_assembler_torch.py:19-37)."""
import torch
from torch.nn import Module


class Assembler(Module):
    """forward(x) = decoder(encoder(x)); x is (snapshot, node ids)."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def forward(self, input_state):
        return self.decoder(self.encoder(input_state))

    def to(self, *args, **kwargs):
        """`assembler(gnn, decoder).to(device)`, the idiom of the reference's scripts
        (examples/torch/citation_prediction.py:81): the decoders' node memories are plain
        tensors, not parameters or buffers, so they are moved through the decoder's own `to`."""
        super().to(*args, **kwargs)
        device = torch._C._nn._parse_to(*args, **kwargs)[0]
        if device is not None:
            for part in (self.encoder, self.decoder):
                if type(part).to is not Module.to:
                    part.to(device)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda" if device is None else device))

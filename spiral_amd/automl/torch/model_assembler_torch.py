"""Encoder + decoder glue (spiro/automl/torch/model_assembler_torch.py:19-37)."""
from torch.nn import Module


class Assembler(Module):
    """forward(x) = decoder(encoder(x)); x is (snapshot, node ids)."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def forward(self, input_state):
        return self.decoder(self.encoder(input_state))

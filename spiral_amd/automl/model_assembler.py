"""`assembler(encoder, decoder)` as in spiro/automl/model_assembler.py:32-34."""
from spiral_amd.automl.torch.model_assembler_torch import Assembler as Assembler


def assembler(encoder, decoder):
    return Assembler(encoder, decoder)

"""Sampling helper of the edge task (spiro/core/dataio.py:29-35)."""
import numpy as np
from numpy import ndarray


def neg_sampling(pos_inds: ndarray, n_items: int, n_samp: int) -> ndarray:
    """n_samp ids in [0, n_items) that avoid the SORTED ids `pos_inds`: draw from the
    shrunken range and shift each draw past the positives below it."""
    pos_inds = np.asarray(pos_inds)
    raw = np.random.randint(0, n_items - len(pos_inds), size=n_samp)
    shifted = pos_inds - np.arange(len(pos_inds))
    return raw + np.searchsorted(shifted, raw, side="right")

"""Host-side pieces of the edge (link-prediction) task."""
import numpy as np

from oracle import oracle as orc
from spiral_amd.core.dataio import neg_sampling


def test_neg_sampling_excludes_positives():
    """tests/core/test_dataio.py:11-15 of the reference."""
    pos = np.sort(np.random.choice(100, 20, replace=False))
    neg = neg_sampling(pos, 100, 50)
    assert len(neg) == 50 and not set(neg.tolist()) & set(pos.tolist())
    assert neg.min() >= 0 and neg.max() < 100
    np.random.seed(3)
    a = neg_sampling(pos, 100, 50)
    np.random.seed(3)
    b = orc.neg_sampling(pos, 100, 50)
    assert np.array_equal(a, b)


def test_process_target_ids():
    pairs = np.array([[7, 2], [2, 9], [7, 7]])
    nodes, pid = orc.process_target_ids(pairs)
    assert nodes.tolist() == [2, 7, 9] and np.array_equal(nodes[pid], pairs)
    ids = np.array([4, 1, 3])
    assert orc.process_target_ids(ids)[0] is ids

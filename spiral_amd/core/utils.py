"""Host-side helpers of the snapshot container (spiro/core/utils.py:24-55)."""
import numpy as np
from numpy import ndarray


def is_sorted(array_to_check: ndarray) -> bool:
    """True when a 1-D array is ascending (ties allowed)."""
    a = np.asarray(array_to_check)
    return bool((a[1:] >= a[:-1]).all())


def get_anchor_position(arr_to_search: ndarray, anchors) -> ndarray:
    """End offset (exclusive, i.e. 1-based last position) of every anchor value in a sorted
    array: [0,0,1,1,2,3,3,3,4,5] with anchors 0..5 -> [2,4,5,8,9,10]."""
    arr = np.asarray(arr_to_search)
    ends = np.searchsorted(arr, np.asarray(list(anchors)), side="left")[1:]
    return np.concatenate([ends, [arr.shape[0]]]).astype(np.int64)


def _get_backend() -> str:
    """The backend in force (spiro/core/utils.py:58-60)."""
    from spiral_amd.core.config import get_backend

    return get_backend()


def printProgressBar(iteration, total, prefix="", suffix="", decimals=1, length=100, fill="█",
                     printEnd="\r") -> None:
    """One line of a terminal progress bar, redrawn in place (spiro/core/utils.py:64-83):
    `prefix |####----| 42.0% suffix`; a newline once `iteration` reaches `total`."""
    done = iteration / float(total)
    cells = int(length * iteration // total)
    print(f"\r{prefix} |{fill * cells}{'-' * (length - cells)}| {100 * done:.{decimals}f}% {suffix}",
          end=printEnd)
    if iteration == total:
        print()

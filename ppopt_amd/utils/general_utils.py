"""Small array helpers used by the host-side mirror of the program classes.

Same names and argument meaning as the reference's ``utils/general_utils.py`` (make_column :9,
make_row :21, select_not_in_list :33, remove_size_zero_matrices :89, num_cpu_cores :97, ppopt_block :117).
"""
import os
from typing import Iterable, List, Sequence, Union

import numpy


def make_column(x: Union[Sequence, numpy.ndarray]) -> numpy.ndarray:
    """Column-vector view/copy of ``x``."""
    return numpy.asarray(x).reshape(-1, 1)


def make_row(x: Union[Sequence, numpy.ndarray]) -> numpy.ndarray:
    """Row-vector view/copy of ``x``."""
    return numpy.asarray(x).reshape(1, -1)


def select_not_in_list(A: numpy.ndarray, coll: Iterable[int]) -> numpy.ndarray:
    """Rows of ``A`` whose index is not in ``coll`` (order preserved)."""
    drop = set(int(i) for i in coll)
    return A[[i for i in range(A.shape[0]) if i not in drop]]


def remove_size_zero_matrices(list_matrices: List[numpy.ndarray]) -> List[numpy.ndarray]:
    """Filters out blocks with an empty dimension."""
    return [m for m in list_matrices if m.shape[0] > 0 and m.shape[1] > 0]


def num_cpu_cores() -> int:
    """Cores this process may run on (affinity mask when the platform has one)."""
    if hasattr(os, 'sched_getaffinity'):
        return len(os.sched_getaffinity(0))
    return os.cpu_count() or 1


def ppopt_block(mat_list) -> numpy.ndarray:
    """``numpy.block`` for a (list of) row(s) of 2-D float blocks, without the generic dispatch overhead."""
    rows = mat_list if isinstance(mat_list[0], list) else [mat_list]
    width = sum(b.shape[1] for b in rows[0])
    height = sum(r[0].shape[0] for r in rows)
    out = numpy.zeros((height, width))
    top = 0
    for r in rows:
        left = 0
        for blk in r:
            h, w = blk.shape
            out[top:top + h, left:left + w] = blk
            left += w
        top += r[0].shape[0]
    return out

"""Constraint manipulation used by presolve and by region assembly.

Host-side restatement (numpy; runs once per program) of the pieces of the reference's
``utils/constraint_utilities.py`` that decide WHICH rows a program has and in WHICH order -- the meaning of every
active-set index on the device depends on it.  Function names follow the reference; line references are to
/root/reference/src/ppopt/utils/constraint_utilities.py.
"""
from typing import List, Optional, Tuple

import numpy

from .general_utils import ppopt_block, select_not_in_list


def constraint_norm(A: numpy.ndarray) -> numpy.ndarray:
    """Row-wise L2 norms as a column (:14-22)."""
    return numpy.linalg.norm(A, axis=1, keepdims=True)


def scale_constraint(A: numpy.ndarray, b: numpy.ndarray) -> List[numpy.ndarray]:
    """Rows of [A | b] divided by ||A_i||_2 (:25-35)."""
    w = 1.0 / numpy.linalg.norm(A, axis=1, keepdims=True)
    return [A * w, b * w]


def detect_implicit_equalities(A: numpy.ndarray, b: numpy.ndarray) -> List[List[int]]:
    """Pairs (i, j), i <= j, of rows of [A | b] with u_i ~ -u_j after normalisation: a two-of-three vote between
    <u_i,u_j> ~ -1 (1e-8), ||u_i - u_j|| ~ 0 (1e-12) and allclose(u_i, -u_j)  (:38-97)."""
    blk = numpy.hstack([A, b.reshape(-1, 1)]).astype(float)
    blk = blk / numpy.linalg.norm(blk, axis=1, keepdims=True)
    blk = blk / numpy.linalg.norm(blk, axis=1, keepdims=True)
    # all pairs at once (the reference loops over them, :70-95): Gram matrix, pairwise distances, element-wise closeness
    with numpy.errstate(invalid='ignore'):
        vote_dot = numpy.abs(blk @ blk.T + 1.0) <= 1e-8
        diff = blk[:, None, :] - blk[None, :, :]
        vote_same = numpy.sqrt(numpy.sum(diff * diff, axis=2)) <= 1e-12
        vote_close = numpy.all(numpy.abs(blk[:, None, :] + blk[None, :, :]) <= 1e-8 + 1e-5 * numpy.abs(blk[None, :, :]), axis=2)
    votes = vote_dot.astype(numpy.int8) + vote_same + vote_close
    hit = numpy.triu(votes >= 2)
    return [[int(i), int(j)] for i, j in numpy.argwhere(hit)]


def remove_zero_rows(A: numpy.ndarray, b: numpy.ndarray) -> List[numpy.ndarray]:
    """Drops rows of A that are exactly zero (:101-111)."""
    keep = [i for i in range(A.shape[0]) if numpy.any(A[i] != 0)]
    return [A[keep], b[keep]]


def remove_duplicate_rows(A: numpy.ndarray, b: numpy.ndarray) -> List[numpy.ndarray]:
    """Keeps the first occurrence of every bit-identical row of [A | b], original order (:125-135)."""
    if A.size == 0 or b.size == 0:
        return [A, b]
    stacked = numpy.hstack((A, b.reshape(b.size, 1)))
    first = numpy.sort(numpy.unique(stacked, axis=0, return_index=True)[1])
    return [A[first], b[first]]


def find_redundant_constraints(A: numpy.ndarray, b: numpy.ndarray, equality_set: Optional[List[int]] = None,
                               solver=None) -> List[int]:
    """Indices of the rows that can be active together with the equalities: row i survives iff
    {A y <= b, rows equality_set and i as equalities} is feasible (:186-200).  ``solver`` is a Solver-like object."""
    eq = list(equality_set or [])
    todo = [i for i in range(A.shape[0]) if i not in eq]
    if hasattr(solver, 'solve_lp_batch'):
        res = solver.solve_lp_batch(None, A, b, [[*eq, i] for i in todo]) if todo else []
    else:
        res = [solver.solve_lp(None, A, b, [*eq, i]) for i in todo]
    dead = {i for i, r in zip(todo, res) if r is None}
    return [i for i in range(A.shape[0]) if i not in dead]


def is_full_rank(A: numpy.ndarray, indices: Optional[List[int]] = None) -> bool:
    """rank(A[indices]) == len(indices); empty selections are full rank (:222-236)."""
    if indices is None:
        return numpy.linalg.matrix_rank(A) == A.shape[0]
    if len(indices) == 0:
        return True
    return numpy.linalg.matrix_rank(A[indices]) == len(indices)


def cheap_remove_redundant_constraints(A: numpy.ndarray, b: numpy.ndarray) -> List[numpy.ndarray]:
    """zero rows out, unit row norms, duplicates out (:239-257)."""
    A, b = remove_zero_rows(A, b)
    A, b = scale_constraint(A, b)
    return remove_duplicate_rows(A, b)


def get_indices_of_zero_rows(A: numpy.ndarray, epsilon: float = 10 ** (-6)) -> Tuple[list, list]:
    """(rows with norm >= epsilon, rows with norm < epsilon)  (:281-288)."""
    norms = numpy.linalg.norm(A, axis=1) if A.size else numpy.zeros(A.shape[0])
    kept = [i for i in range(A.shape[0]) if norms[i] >= epsilon]
    gone = [i for i in range(A.shape[0]) if not norms[i] >= epsilon]
    return kept, gone


def shuffle_processed_constraints(A, b, F, A_t, b_t, kept: list, remove: list):
    """Moves the rows ``remove`` of the main block (which do not involve x) into the parametric block as
    -F theta <= b (:291-317)."""
    if len(remove) > 0:
        A_t = ppopt_block([[A_t], [-F[remove]]])
        b_t = ppopt_block([[b_t], [b[remove]]])
    return A[kept], b[kept], F[kept], A_t, b_t


def get_independent_rows(A: numpy.ndarray) -> List[int]:
    """Indices where the rank of the leading rows grows (:320-332); the last row is never examined, as in the
    reference (its loop stops one short)."""
    m = A.shape[0]
    ranks = numpy.zeros(m)
    for i in range(m - 1):
        ranks[i] = numpy.linalg.matrix_rank(A[:i + 1])
    grew = numpy.diff(ranks, prepend=0) > 0
    return [i for i, g in enumerate(grew) if g]


def generate_reduced_equality_constraints(A, b, F, equality_indices):
    """Drops linearly dependent equality rows (:335-362)."""
    if len(equality_indices) == 0:
        return A, b, F, []
    if is_full_rank(A, equality_indices):
        return A, b, F, equality_indices
    keep = get_independent_rows(A[equality_indices])
    A_in, b_in, F_in = (select_not_in_list(M, equality_indices) for M in (A, b, F))
    return numpy.vstack([A[keep], A_in]), numpy.vstack([b[keep], b_in]), numpy.vstack([F[keep], F_in]), keep


def process_program_constraints(A, b, F, A_t, b_t, epsilon: float = 10 ** (-6)):
    """Rows with ||[A_i | -F_i]|| < eps, then rows with ||A_i|| < eps, move to the parametric block (:365-401)."""
    keep, move = get_indices_of_zero_rows(ppopt_block([[A, -F]]), epsilon)
    A, b, F, A_t, b_t = shuffle_processed_constraints(A, b, F, A_t, b_t, keep, move)
    keep, move = get_indices_of_zero_rows(A, epsilon)
    return shuffle_processed_constraints(A, b, F, A_t, b_t, keep, move)


def find_implicit_equalities(A, b, F, equality_indices):
    """Inequality pairs L <= a'x - f'theta <= L become one equality; equalities are moved to the top (:404-466)."""
    pairs = detect_implicit_equalities(ppopt_block([[A, -F]]), ppopt_block([[b]]))
    keep = sorted({p[0] for p in pairs})
    drop = [i for i in sorted({p[1] for p in pairs}) if i not in keep]
    top = [*equality_indices, *keep]
    rest = [i for i in range(A.shape[0]) if i not in top and i not in drop]
    A = ppopt_block([[A[top]], [A[rest]]])
    b = ppopt_block([[b[top]], [b[rest]]])
    F = ppopt_block([[F[top]], [F[rest]]])
    return A, b, F, list(range(len(top)))


def numerically_nonzero_rows(A) -> List[int]:
    """Rows with some |entry| > 1e-8 (:469-470)."""
    return [i for i in range(A.shape[0]) if not numpy.allclose(A[i], 0, atol=10 ** -8)]


def remove_numerically_zero_rows(A, b) -> Tuple[numpy.ndarray, numpy.ndarray]:
    keep = numerically_nonzero_rows(A)
    return A[keep], b[keep]

"""Chebyshev ball of a polytope {x: Ax <= b} through the LP plug (reference: utils/chebyshev_ball.py:10-63)."""
from typing import Optional, Sequence

import numpy

from .constraint_utilities import constraint_norm


def chebyshev_ball(A: numpy.ndarray, b: numpy.ndarray, equality_constraints: Optional[Sequence[int]] = None,
                   solver=None):
    """max r s.t. A x + ||A_i|| r <= b (norm 0 on equality rows), r >= 0.  Returns the SolverOutput
    (sol = [x, r]) or None when the LP is infeasible or unbounded."""
    from ..solver import Solver
    solver = solver or Solver()
    eq = list(equality_constraints or [])
    n = A.shape[1]
    norms = constraint_norm(A)
    norms[eq] = 0.0
    c = numpy.zeros((n + 1, 1))
    c[n, 0] = -1.0
    A_ball = numpy.vstack([numpy.hstack([A, norms]), c.T])
    b_ball = numpy.vstack([b.reshape(-1, 1), numpy.zeros((1, 1))])
    return solver.solve_lp(c, A_ball, b_ball, eq)

"""MPMIQP_Program: a multiparametric QP some of whose variables are binary (reference: mpmiqp_program.py:11-148).

    min_{x,y}  1/2 [x,y]' Q [x,y] + theta' H' [x,y] + c' [x,y] + c_c + c_t' theta + 1/2 theta' Q_t theta

Feasibility questions do not involve the objective, so presolve and the binary tree are those of MPMILP_Program
(device-batched LPs); only the substitution of a fixation carries the quadratic terms.
"""
from typing import List, Optional, Union

import numpy

from .mpmilp_program import MPMILP_Program
from .mpqp_program import MPQP_Program
from .solver import Solver


class MPMIQP_Program(MPMILP_Program):
    def __init__(self, A, b, c, H, Q, A_t, b_t, F, binary_indices: List, c_c=None, c_t=None, Q_t=None,
                 equality_indices=None, solver: Optional[Solver] = None, post_process: bool = True):
        self.Q = numpy.asarray(Q, dtype=numpy.float64)
        super().__init__(A, b, c, H, A_t, b_t, F, binary_indices, c_c, c_t, Q_t, equality_indices, solver,
                         post_process=False)
        if post_process:
            self.post_process()

    def evaluate_objective(self, x: numpy.ndarray, theta_point: numpy.ndarray) -> float:
        v = 0.5 * x.T @ self.Q @ x + theta_point.T @ self.H.T @ x + self.c.T @ x + self.c_c \
            + self.c_t.T @ theta_point + 0.5 * theta_point.T @ self.Q_t @ theta_point
        return float(v[0, 0])

    def generate_substituted_problem(self, fixed_combination: Union[numpy.ndarray, List[int]], deferred: bool = False):
        """The continuous mpQP with the binaries fixed (mpmiqp_program.py:70-115): the binary block of Q moves into
        the constant, the mixed block into the linear term.  ``deferred`` (the enumeration's batch construction): the presolve's
        redundancy LPs are left to the caller (``_redundancy_request`` / ``_redundancy_apply``) and the diagnostic LPs are not posed."""
        A_cont, b, F, eq, y = self._substituted_rows(fixed_combination)
        ci, bi = self.cont_indices, self.binary_indices
        Q_c = self.Q[:, ci][ci]
        Q_d = self.Q[:, bi][bi]
        Q_mix = self.Q[:, ci][bi]
        c = self.c[ci] + Q_mix.T @ y
        c_c = self.c_c + self.c[bi].T @ y + 0.5 * y.T @ Q_d @ y
        H_c = self.H[ci]
        H_d = self.H[bi]
        c_t = self.c_t + (y.T @ H_d).T
        return MPQP_Program(A_cont, b, c, H_c, Q_c, self.A_t, self.b_t, F, c_c, c_t, self.Q_t, eq, self.solver,
                            post_process=not deferred, _diagnostics=not deferred)

    def generate_relaxed_problem(self, process: bool = True) -> MPQP_Program:
        A, b, F = self._relaxation_rows()
        return MPQP_Program(A, b, self.c, self.H, self.Q, self.A_t, self.b_t, F, self.c_c, self.c_t, self.Q_t,
                            self.equality_indices, self.solver, post_process=process)

    def solve_theta(self, theta_point: numpy.ndarray):
        """The MIQP at a fixed theta needs a QP backend, which is outside the combinatorial path."""
        raise NotImplementedError('solve_theta of a mixed-integer QP needs a QP solver; only LPs run on the device')

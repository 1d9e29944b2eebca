"""MPQP_Program: the quadratic program class (reference: mpqp_program.py:15-322).

    min_x 1/2 x'Qx + theta'H'x + c'x   s.t.  A x <= b + F theta,  A_eq x = b_eq + F_eq theta,  A_t theta <= b_t
"""
from typing import List, Optional, Tuple

import numpy

from .mplp_program import MPLP_Program
from .solver import SolverOutput


class MPQP_Program(MPLP_Program):
    def __init__(self, A, b, c, H, Q, A_t, b_t, F, c_c: Optional[numpy.ndarray] = None,
                 c_t: Optional[numpy.ndarray] = None, Q_t: Optional[numpy.ndarray] = None, equality_indices=None,
                 solver=None, post_process=True, _diagnostics=True):
        self.Q = numpy.asarray(Q).astype('float64')
        super().__init__(A, b, c, H, A_t, b_t, F, c_c, c_t, Q_t, equality_indices, solver, post_process=False, _diagnostics=_diagnostics)
        if post_process:
            self.post_process()

    def evaluate_objective(self, x, theta_point) -> float:
        v = 0.5 * x.T @ self.Q @ x + theta_point.T @ self.H.T @ x + self.c.T @ x + self.c_c \
            + self.c_t.T @ theta_point + 0.5 * theta_point.T @ self.Q_t @ theta_point
        return float(v[0, 0])

    def warnings(self) -> List[str]:
        out = MPLP_Program.warnings(self)
        if self.Q.shape[0] != self.Q.shape[1]:
            out.append(f'Q matrix is not square with dimensions {self.Q.shape}')
        if self.Q.shape[0] != self.A.shape[1] or self.Q.shape[1] != self.A.shape[1]:
            out.append('Dimensions of Q and A matrices disagree in number of x parameters')
        if self.Q.shape[0] == self.Q.shape[1]:
            ev = numpy.linalg.eigvals(self.Q)
            if min(ev) < 0:
                out.append(f'Non-convex quadratic program detected, with eigenvalues {ev}')
            elif min(ev) < 10 ** -4:
                out.append(f'Possible positive semi-definite nature detected in Q, eigenvalues {ev}')
        return out

    def optimal_control_law(self, active_set: List[int]) -> Tuple:
        """Host evaluation of the KKT system [[A_as, 0], [Q, A_as']] (mpqp_program.py:146-198); the batched device
        equivalent lives in csrc/kkt.hpp.  Raises numpy.linalg.LinAlgError on an exactly singular matrix."""
        A_hat = self.A[active_set]
        k = len(active_set)
        M = numpy.block([[A_hat, numpy.zeros((k, k))], [self.Q, A_hat.T]])
        consts = numpy.linalg.solve(M, numpy.vstack([self.b[active_set], -self.c]))
        mats = numpy.linalg.solve(M, numpy.vstack([self.F[active_set], -self.H]))
        nx = self.num_x()
        return mats[:nx], consts[:nx], mats[nx:], consts[nx:]

    def solve_theta(self, theta_point: numpy.ndarray) -> Optional[SolverOutput]:
        """The QP at a fixed theta (mpqp_program.py:109-143): SolverOutput (obj, sol, slack, active_set, dual) or None when
        theta violates A_t theta <= b_t or the QP is infeasible there.  Solved on the device (``solve_theta_batch``)."""
        th = numpy.asarray(theta_point, dtype=float).reshape(-1, 1)
        if not numpy.all(self.A_t @ th <= self.b_t):
            return None
        return self.solve_theta_batch(th.reshape(1, -1))[0]

    def solve_theta_batch(self, theta_points: numpy.ndarray) -> List[Optional[SolverOutput]]:
        """``solve_theta`` for many parameter points in one device launch (theta_points [m, n_theta]): the KKT conditions as a
        linear complementarity problem in the multipliers, Lemke's method, one wavefront per point (csrc/qp.hpp).  Needs a
        positive definite Q.  Points outside A_t theta <= b_t are solved like any other (the single-point form filters them)."""
        th = numpy.ascontiguousarray(theta_points, dtype=numpy.float64).reshape(-1, self.num_t())
        status, x, lam, act = self.engine().qp_solve_batch(th)
        out: List[Optional[SolverOutput]] = []
        for p in range(len(th)):
            if status[p] != 0:
                out.append(None)
                continue
            tp, xp = th[p].reshape(-1, 1), x[p].reshape(-1, 1)
            slack = (self.b + self.F @ tp - self.A @ xp).ravel()
            out.append(SolverOutput(self.evaluate_objective(xp, tp), x[p].copy(), slack, numpy.flatnonzero(act[p]), lam[p].copy()))
        return out

"""MPLP_Program: host-side mirror of the reference program class for the combinatorial path.

    min_x  theta' H' x + c' x   s.t.  A x <= b + F theta  (rows ``equality_indices`` as equalities),  A_t theta <= b_t

Same constructor signature, the same presolve (and therefore the same row indexing) and the same per-active-set
methods as the reference (mplp_program.py:27-678).  What differs is where the work happens: every LP goes through
the deterministic-solver plug, whose only product backend is the MI355X (ppopt_amd.solver.Solver), and
``check_feasibility`` / ``check_optimality`` / ``gen_cr_from_active_set`` run on the device through the C ABI.
"""
import warnings
from typing import List, Optional, Tuple

import numpy

from .solver import Solver
from .utils.chebyshev_ball import chebyshev_ball
from .utils.constraint_utilities import (constraint_norm, find_implicit_equalities, find_redundant_constraints,
                                         generate_reduced_equality_constraints, is_full_rank,
                                         process_program_constraints)
from .utils.general_utils import make_column, ppopt_block, select_not_in_list


class MPLP_Program:
    def __init__(self, A, b, c, H, A_t, b_t, F, c_c=None, c_t=None, Q_t=None, equality_indices=None, solver=None,
                 post_process=True, _diagnostics=True):
        # _diagnostics=False (private; the mixed-integer enumeration's batch construction): the two LPs behind ``warnings()`` -- whose only
        # outcome is a UserWarning, which the enumeration discards -- are not posed
        self.A, self.b, self.c, self.H = A, b, c, H
        self.A_t, self.b_t, self.F = A_t, b_t, F
        self.c_c = numpy.array([[0.0]]) if c_c is None else c_c
        self.c_t = numpy.zeros((self.num_t(), 1)) if c_t is None else c_t
        self.Q_t = numpy.zeros((self.num_t(), self.num_t())) if Q_t is None else Q_t
        self.equality_indices = list(equality_indices) if equality_indices is not None and len(equality_indices) else []
        self.solver = Solver() if solver is None else solver
        self._engine = None
        self._engine_closed = None
        self._closing_rows = None

        self.base_constraint_processing()
        if _diagnostics:
            for msg in self.warnings():
                warnings.warn(msg, UserWarning)
        if post_process:
            self.post_process()

    def _rows_changed(self) -> None:
        """The rows have changed: both device handles (the program's own and the one with the closed parameter set) describe the old
        rows and are given back; the next ``engine()`` call builds them again."""
        if getattr(self, '_engine', None) is not None:
            self._engine.close()
        self._engine = None
        if getattr(self, '_engine_closed', None) is not None:
            self._engine_closed.close()
        self._engine_closed = None
        self._closing_rows = None

    # ---- presolve (mplp_program.py:110-134, 276-322) -------------------------------------------------------
    def base_constraint_processing(self):
        if len(self.equality_indices) != 0:
            eq = self.equality_indices
            self.A = numpy.vstack([self.A[eq], select_not_in_list(self.A, eq)])
            self.b = numpy.vstack([self.b[eq], select_not_in_list(self.b, eq)])
            self.F = numpy.vstack([self.F[eq], select_not_in_list(self.F, eq)])
            self.equality_indices = list(range(len(eq)))
        self.constraint_datatype_conversion()
        self.A, self.b, self.F, self.A_t, self.b_t = process_program_constraints(self.A, self.b, self.F, self.A_t,
                                                                                 self.b_t)
        self.scale_constraints()
        self.A, self.b, self.F, self.equality_indices = find_implicit_equalities(self.A, self.b, self.F,
                                                                                 self.equality_indices)
        self.A, self.b, self.F, self.equality_indices = generate_reduced_equality_constraints(
            self.A, self.b, self.F, self.equality_indices)
        self._rows_changed()

    def post_process(self):
        self.process_constraints()

    def constraint_datatype_conversion(self) -> None:
        for name in ('A', 'c', 'b', 'F', 'A_t', 'b_t', 'H', 'c_c', 'c_t', 'Q_t'):
            setattr(self, name, numpy.asarray(getattr(self, name)).astype('float64'))

    def scale_constraints(self) -> None:
        """Rows of [A | -F] (and b) to unit L2 norm."""
        norm = constraint_norm(numpy.hstack([self.A, -self.F]))
        self.A, self.b, self.F = self.A / norm, self.b / norm, self.F / norm
        self._rows_changed()

    def _redundancy_request(self):
        """(PA, Pb, equality sets) of ``process_constraints``' LPs -- one per non-equality row of [[A, -F], [0, A_t]], that row and the
        program's equalities as equalities -- without solving them (the enumeration poses those of all sub-programs together)."""
        PA = ppopt_block([[self.A, -self.F], [numpy.zeros((self.A_t.shape[0], self.A.shape[1])), self.A_t]])
        Pb = ppopt_block([[self.b], [self.b_t]])
        eq = list(self.equality_indices)
        todo = [i for i in range(PA.shape[0]) if i not in eq]
        return PA, Pb, eq, todo

    def _redundancy_apply(self, request, feasible) -> None:
        """Second half of ``process_constraints``: ``feasible[j]`` says whether row ``todo[j]`` of the request can be active."""
        PA, _, _, todo = request
        dead = {i for i, ok in zip(todo, feasible) if not ok}
        self._keep_rows([i for i in range(PA.shape[0]) if i not in dead])

    def process_constraints(self) -> None:
        """Removes rows that cannot be active: one LP per non-equality row of [[A, -F], [0, A_t]]."""
        PA = ppopt_block([[self.A, -self.F], [numpy.zeros((self.A_t.shape[0], self.A.shape[1])), self.A_t]])
        Pb = ppopt_block([[self.b], [self.b_t]])
        self._keep_rows(find_redundant_constraints(PA, Pb, self.equality_indices, solver=self.solver))

    def _keep_rows(self, saved) -> None:
        n_c = self.num_constraints()
        upper = [i for i in saved if i < n_c]
        lower = [i - n_c for i in saved if i >= n_c]
        self.A, self.F, self.b = self.A[upper], self.F[upper], self.b[upper]
        self.A_t, self.b_t = self.A_t[lower], self.b_t[lower]
        self._rows_changed()

    # ---- sizes ------------------------------------------------------------------------------------------------
    def num_x(self) -> int:
        return self.A.shape[1]

    def num_t(self) -> int:
        return self.F.shape[1]

    def num_constraints(self) -> int:
        return self.A.shape[0]

    def num_inequality_constraints(self) -> int:
        return self.A.shape[0] - len(self.equality_indices)

    def num_equality_constraints(self) -> int:
        return len(self.equality_indices)

    def evaluate_objective(self, x: numpy.ndarray, theta_point: numpy.ndarray) -> float:
        v = theta_point.T @ self.H.T @ x + self.c.T @ x + self.c_c + self.c_t.T @ theta_point \
            + 0.5 * theta_point.T @ self.Q_t @ theta_point
        return float(v[0, 0])

    # ---- diagnostics (mplp_program.py:162-218) ------------------------------------------------------------------
    def warnings(self) -> List[str]:
        out = []
        if self.b.ndim != 2:
            out.append(f'The b matrix is not a column vector b{self.b.shape}')
            self.b = make_column(self.b)
            out.append('This has been corrected')
        if self.c.ndim != 2:
            out.append(f'The c vector is not a column vector c{self.c.shape}')
            self.c = make_column(self.c)
            out.append('This has been corrected')
        if self.A.shape[1] != self.c.shape[0]:
            out.append(f'The A and b matrices disagree in number of parameters A{self.A.shape}, c{self.c.shape}')
        if self.A.shape[0] != self.b.shape[0]:
            out.append(f'The A and b matrices disagree in vertical dimension A{self.A.shape}, b{self.b.shape}')
        if self.A_t.shape[0] != self.b_t.shape[0]:
            out.append(f'The A and b matrices disagree in vertical dimension A{self.A_t.shape}, b{self.b_t.shape}')
        if self.A.shape[0] != self.F.shape[0]:
            out.append(f'The A and F matrices disagree in vertical dimension A{self.A.shape}, F {self.F.shape}')
        if self.F.shape[1] != self.A_t.shape[1]:
            out.append(f'The F and A_t matrices disagree in dimension A_t {self.A_t.shape}, F {self.F.shape}, '
                       f'inconsistent number of parameters')
        if not out:
            if self.feasible_space_chebychev_ball() is None:
                out.append('The chebychev ball has either a radius of zero, or the problem is not feasible!')
            if not self._lp_feasible(self.equality_indices):
                out.append('The multiparametric program, as stated, is not feasible!')
        return out

    def feasible_space_chebychev_ball(self):
        PA = numpy.vstack([numpy.hstack([self.A, -self.F]),
                           numpy.hstack([numpy.zeros((self.A_t.shape[0], self.num_x())), self.A_t])])
        Pb = numpy.vstack([self.b, self.b_t])
        return chebyshev_ball(PA, Pb, equality_constraints=self.equality_indices, solver=self.solver)

    def feasible_theta_point(self) -> Optional[numpy.ndarray]:
        sol = self.feasible_space_chebychev_ball()
        return None if sol is None else sol.sol[self.num_x():self.num_x() + self.num_t()].reshape(-1, 1)

    def valid_parameter_realization(self, theta_point) -> bool:
        return bool(numpy.all(self.A_t @ theta_point <= self.b_t))

    def solve_theta(self, theta_point: numpy.ndarray):
        """The LP at a fixed theta (mplp_program.py:324-352)."""
        if not self.valid_parameter_realization(theta_point):
            return None
        sol = self.solver.solve_lp(self.H @ theta_point + self.c, self.A, self.b + self.F @ theta_point,
                                   self.equality_indices)
        if sol is not None:
            sol.obj += float((self.c_c + self.c_t.T @ theta_point + 0.5 * theta_point.T @ self.Q_t @ theta_point)[0, 0])
        return sol

    def display_warnings(self) -> None:
        print(self.warnings())

    def solve_theta_variable(self):
        """theta left as a variable: min c'x s.t. [A | -F][x; theta] <= b (mplp_program.py:355-370)."""
        c_prime = numpy.vstack([self.c, numpy.zeros((self.num_t(), 1))])
        return self.solver.solve_lp(c_prime, numpy.hstack([self.A, -self.F]), self.b, self.equality_indices)

    def solve_theta_batch(self, theta_points: numpy.ndarray):
        """``solve_theta`` for many parameter points in one device launch (theta_points [m, n_theta]); entries are
        SolverOutput or None, in order."""
        from . import _lib
        th = numpy.ascontiguousarray(theta_points, dtype=numpy.float64).reshape(-1, self.num_t())
        inside = numpy.all(th @ self.A_t.T <= self.b_t.reshape(1, -1), axis=1)
        out = [None] * len(th)
        if not inside.any():
            return out
        pts = th[inside]
        b = self.b.reshape(1, -1) + pts @ self.F.T                       # [m, n_c]
        c = self.c.reshape(1, -1) + pts @ self.H.T                       # [m, n_x]
        flags = numpy.zeros((len(pts), self.num_constraints()), dtype=numpy.uint8)
        flags[:, list(self.equality_indices)] = 1
        status, x, obj, _ = _lib.lp_solve_batch(self.A, b, c, flags, device=self.solver.device)
        from .solver import SolverOutput
        for j, i in enumerate(numpy.flatnonzero(inside)):
            if status[j] != _lib.LP_OPTIMAL:
                continue
            t = pts[j].reshape(-1, 1)
            slack = b[j] - self.A @ x[j]
            const = float((self.c_c + self.c_t.T @ t + 0.5 * t.T @ self.Q_t @ t)[0, 0])
            out[i] = SolverOutput(float(obj[j]) + const, x[j].copy(), slack, numpy.flatnonzero(numpy.abs(slack) <= 1e-10), None)
        return out

    def gen_optimal_active_set(self) -> Optional[List[int]]:
        """An optimal active set found by sampling around the Chebyshev centre of the feasible space
        (mplp_program.py:588-618); the up to 500 sample LPs are one device batch instead of a loop."""
        sol = self.feasible_space_chebychev_ball()
        if sol is None:
            return None
        prng = numpy.random.default_rng()
        centre = sol.sol[self.num_x():self.num_x() + self.num_t()].reshape(1, -1)
        radius = float(sol.sol[-1])
        samples = centre + prng.uniform(-radius, radius, (500, self.num_t()))
        for res in self.solve_theta_batch(samples):
            if res is not None and res.active_set.size <= self.num_x():
                return res.active_set.tolist()
        return None

    def sample_theta_space(self, num_samples: int = 100) -> Optional[list]:
        """Random walk through the feasible parameter space collecting the active sets met (mplp_program.py:632-664);
        every step depends on the previous one, so the LPs are solved one at a time."""
        sol = self.feasible_space_chebychev_ball()
        if sol is None:
            return None
        prng = numpy.random.default_rng()
        theta = sol.sol[self.num_x():self.num_x() + self.num_t()].reshape(-1, 1)
        radius = float(sol.sol[-1])
        found = []
        for _ in range(num_samples):
            step = prng.standard_normal(self.num_t()).reshape(-1, 1)
            new_theta = theta + prng.random() * radius * step / numpy.linalg.norm(step, 2)
            res = self.solve_theta(new_theta)
            if res is not None:
                found.append(tuple(res.active_set.tolist()))
                theta = new_theta
        return [list(a) for a in set(found)]

    # ---- per-active-set primitives ----------------------------------------------------------------------------------
    def _lp_feasible(self, active_set) -> bool:
        PA = ppopt_block([[self.A, -self.F], [numpy.zeros((self.A_t.shape[0], self.num_x())), self.A_t]])
        Pb = ppopt_block([[self.b], [self.b_t]])
        return self.solver.solve_lp(numpy.zeros((self.num_x() + self.num_t(), 1)), PA, Pb, list(active_set)) is not None

    def engine(self, device: Optional[int] = None, closed: bool = False):
        """The device-resident twin of this program (created on first use; dropped when the rows change).  Default
        device: the one the program's solver runs its presolve LPs on (``Solver(device=...)``)."""
        from . import _lib
        if device is None:
            device = int(getattr(self.solver, 'device', 0) or 0)
        if closed:
            # the combinatorial drivers: a parameter set without a vertex is closed for the device (_engine_parameter_rows); the handle
            # is a second one, the other drivers (graph, geometric, point location) keep the program's own rows.  (Both handles are
            # dropped whenever the rows change: _rows_changed.)
            eng_c = getattr(self, '_engine_closed', None)
            if eng_c is not None and eng_c.device == device:
                return eng_c
            A_t, b_t = self._engine_parameter_rows()
            if A_t.shape[0] != self.A_t.shape[0]:
                if eng_c is not None:
                    eng_c.close()
                self._engine_closed = _lib.Engine(self.A, self.b, self.F, self.c, self.H, getattr(self, 'Q', None), A_t, b_t,
                                                  len(self.equality_indices), device=device)
                self._engine_closed.n_tc_program = int(self.A_t.shape[0])   # rows beyond it are the closing rows
                return self._engine_closed
        if self._engine is None or self._engine.device != device:
            Q = getattr(self, 'Q', None)
            self._engine = _lib.Engine(self.A, self.b, self.F, self.c, self.H, Q, self.A_t, self.b_t,
                                       len(self.equality_indices), device=device)
        return self._engine

    CLOSING_BOX_LIMIT = 1e3     # largest |theta| of the program's parameter box for which closing rows are used (see below)

    def _engine_parameter_rows(self):
        """(A_t, b_t) as the device handle gets them.  The register-resident kernels start their theta-space LPs at a VERTEX of the
        parameter set {A_t theta <= b_t}.  The presolve may leave a set without one (fewer than n_theta independent rows: the others
        were implied by rows in x -- the 51-region variant of the double-integrator MPC keeps two parallel rows); such a program
        would run on the slower LDS-engine kernels.  Then the box of theta over the whole program {A x <= b + F theta, A_t theta
        <= b_t} (2 n_theta LPs through the solver), widened by half its size, is appended as closing rows.  Every parameter any
        active set can be feasible at lies strictly inside that box, so the rows are strictly redundant in every LP the path poses
        (optimality, full dimension, facets): verdicts and regions do not change, and no region can list one of them -- the solve
        checks that (mpqp_hip_combinatorial._closing_rows_unused) and repeats without them otherwise.  Only for boxes of moderate
        size (CLOSING_BOX_LIMIT): a program that keeps big-M rows has a box of 1e8, and rows that far out ruin the absolute
        tolerances of the LPs they take part in (measured: facet lists of regions change).  MPC_NO_THETA_CLOSE=1: off."""
        import os
        A_t, b_t = self.A_t, self.b_t
        nt = self.num_t()
        if os.environ.get('MPC_NO_THETA_CLOSE', '0') == '1':
            return A_t, b_t
        key = (id(self.A), id(self.b), id(self.F), id(A_t), id(b_t), self.A.shape, A_t.shape)
        cached = getattr(self, '_closing_rows', None)      # the decision is made once per set of rows (2 n_theta LPs when a set has no vertex)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        out = self._closing_rows_compute(A_t, b_t, nt)
        self._closing_rows = (key, out[0], out[1], (self.A, self.b, self.F, A_t, b_t))      # (the arrays are held: their ids stay theirs)
        return out

    def _closing_rows_compute(self, A_t, b_t, nt):
        if A_t.shape[0] >= nt and numpy.linalg.matrix_rank(A_t) >= nt:
            return A_t, b_t
        nx = self.num_x()
        PA = ppopt_block([[self.A, -self.F], [numpy.zeros((A_t.shape[0], nx)), A_t]])
        Pb = ppopt_block([[self.b], [b_t]])
        lo, hi = numpy.zeros(nt), numpy.zeros(nt)
        for j in range(nt):
            for sign, out in ((1.0, lo), (-1.0, hi)):
                c = numpy.zeros((nx + nt, 1))
                c[nx + j, 0] = sign
                res = self.solver.solve_lp(c, PA, Pb, list(self.equality_indices))
                if res is None:
                    return A_t, b_t        # unbounded (or infeasible) in this direction: no closing rows, the program keeps the slower kernels
                out[j] = sign * res.obj
        if max(float(numpy.max(numpy.abs(lo))), float(numpy.max(numpy.abs(hi)))) > self.CLOSING_BOX_LIMIT:
            return A_t, b_t
        pad = 0.5 * (hi - lo) + 1.0
        rows = numpy.vstack([numpy.eye(nt), -numpy.eye(nt)])
        rhs = numpy.concatenate([hi + pad, -(lo - pad)]).reshape(-1, 1)
        return numpy.vstack([A_t, rows]), numpy.vstack([b_t.reshape(-1, 1), rhs])

    def release_engine(self) -> None:
        """Gives the device handle back (its blocks return to the library's pool); a later call re-creates it."""
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        if getattr(self, '_engine_closed', None) is not None:
            self._engine_closed.close()
            self._engine_closed = None

    def _device_status(self, active_set) -> int:
        eng = self.engine()
        cand = numpy.asarray(list(active_set), dtype=numpy.int32).reshape(1, -1)
        status, *_ = eng.check_level(cand, numpy.zeros((0, 2), dtype=numpy.uint64), False)
        return int(status[0])

    def check_active_set_rank(self, active_set: List[int]) -> bool:
        return is_full_rank(self.A, list(active_set))

    def check_feasibility(self, active_set: List[int], check_rank=True) -> bool:
        """Rank test + feasibility of {Ax <= b + F theta, A_t theta <= b_t, rows active_set active}
        (mplp_program.py:411-444).  With the rank test it is the device verdict; without it, the LP alone."""
        if not check_rank:
            return self._lp_feasible(active_set)
        from . import _lib
        return self._device_status(active_set) != _lib.INFEASIBLE

    def check_optimality(self, active_set) -> bool:
        """True when some theta makes the active set optimal (mplp_program.py:446-569, mpqp_program.py:203-322; the
        reference returns a dict or None and the drivers only use its truth value)."""
        from . import _lib
        return self._device_status(active_set) in (_lib.OPTIMAL_NO_REGION, _lib.REGION)

    def optimal_control_law(self, active_set: List[int]) -> Tuple:
        """(A_x, b_x, A_l, b_l) via the pseudo-inverse of the active rows (mplp_program.py:372-395)."""
        aux = numpy.linalg.pinv(self.A[active_set])
        return aux @ self.F[active_set], aux @ self.b[active_set], -aux.T @ self.H, -aux.T @ self.c

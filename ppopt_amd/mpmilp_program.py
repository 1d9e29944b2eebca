"""MPMILP_Program: a multiparametric LP some of whose variables are binary (reference: mpmilp_program.py:12-269).

    min_{x,y}  theta' H' [x,y] + c' [x,y] + c_c + c_t' theta + 1/2 theta' Q_t theta
    s.t.       A [x,y] <= b + F theta  (rows ``equality_indices`` as equalities),  A_t theta <= b_t,  y binary

Same constructor, presolve and methods as the reference.  The reference hands its mixed-integer feasibility
questions to Gurobi one at a time (process_constraints: one MILP per row, mpmilp_program.py:129-137;
check_bin_feasibility: one MILP per tree node, :203-237).  Here every such question is the same thing -- "does the LP
of some full binary fixation have a solution" -- and the LPs of the fixations (times the rows, for the presolve) go
to the MI355X as ``mpc_lp_solve_batch`` launches of bounded size (ppopt_amd.solver.Solver.milp_leaf_feasibility /
milp_any_feasible; beyond 10 binaries whole subtrees are pruned through their LP relaxation); partial fixations are then
answered from the cached table of leaf verdicts without further device work.
"""
from typing import List, Optional

import numpy

from .mplp_program import MPLP_Program
from .solver import Solver, SolverOutput
from .utils.constraint_utilities import detect_implicit_equalities
from .utils.general_utils import ppopt_block


class MPMILP_Program(MPLP_Program):
    def __init__(self, A, b, c, H, A_t, b_t, F, binary_indices=None, c_c=None, c_t=None, Q_t=None,
                 equality_indices=None, solver: Optional[Solver] = None, post_process=True):
        self.binary_indices = list(binary_indices) if binary_indices is not None else []
        self._leaf_table = None
        super().__init__(A, b, c, H, A_t, b_t, F, c_c, c_t, Q_t, equality_indices, solver, post_process=False)
        self.cont_indices = [i for i in range(self.num_x()) if i not in self.binary_indices]
        if len(self.cont_indices) == 0:
            print('Pure Integer case is not considered here only the Mixed case!!!')
        if post_process:
            self.post_process()

    def post_process(self):
        self.process_constraints()

    # ---- presolve (mpmilp_program.py:73-143) ------------------------------------------------------------------------
    def _stacked_constraints(self):
        PA = ppopt_block([[self.A, -self.F], [numpy.zeros((self.A_t.shape[0], self.A.shape[1])), self.A_t]])
        Pb = ppopt_block([[self.b], [self.b_t]])
        return PA, Pb

    def process_constraints(self, find_implicit_equalities=True) -> None:
        """Moves implicit equalities (pairs of opposite inequalities) to the top and removes every inequality row
        that cannot be active for any binary fixation: row i survives iff the mixed-integer system with row i as an
        equality is feasible.  All (row, fixation) LPs are one device launch."""
        self.constraint_datatype_conversion()
        self.scale_constraints()
        self._leaf_table = None

        if find_implicit_equalities:
            pairs = detect_implicit_equalities(ppopt_block([[self.A, -self.F]]), ppopt_block([[self.b]]))
            keep = sorted(set(p[0] for p in pairs))
            remove = [i for i in sorted(set(p[1] for p in pairs)) if i not in keep]
            new_eq = [*self.equality_indices, *keep]
            rest = [i for i in range(self.num_constraints()) if i not in new_eq and i not in remove]
            self.A = ppopt_block([[self.A[new_eq]], [self.A[rest]]])
            self.b = ppopt_block([[self.b[new_eq]], [self.b[rest]]])
            self.F = ppopt_block([[self.F[new_eq]], [self.F[rest]]])
            self.equality_indices = list(range(len(new_eq)))

        PA, Pb = self._stacked_constraints()
        n_eq = self.num_equality_constraints()
        rows = [i + n_eq for i in range(self.num_inequality_constraints())]
        saved = []
        if rows:
            # a fixation under which the program itself is infeasible stays infeasible with one more row tightened: only
            # the feasible leaves are candidates, and a row is settled by the first leaf that admits it
            leaves = numpy.flatnonzero(self.solver.milp_leaf_feasibility(PA, Pb, self.equality_indices, self.binary_indices))
            feasible = self.solver.milp_any_feasible(PA, Pb, [[*self.equality_indices, r] for r in rows],
                                                     self.binary_indices, leaves)
            saved = [r for r, ok in zip(rows, feasible) if ok]
        upper = [*self.equality_indices, *[i for i in saved if i < self.A.shape[0]]]
        self.A, self.F, self.b = self.A[upper], self.F[upper], self.b[upper]
        self._rows_changed()
        self._leaf_table = None

    # ---- binary fixations ------------------------------------------------------------------------------------------------
    def leaf_feasibility(self) -> numpy.ndarray:
        """bool[2^n_bin]: is the LP over (x_cont, theta) feasible with the binaries fixed to that combination
        (row order of Solver.binary_fixations).  Device batches of LPs (Solver.milp_leaf_feasibility: all leaves at once
        for few binaries, level by level with relaxation pruning for many), cached until the constraints change."""
        if self._leaf_table is None:
            PA, Pb = self._stacked_constraints()
            self._leaf_table = self.solver.milp_leaf_feasibility(PA, Pb, list(self.equality_indices), self.binary_indices)
        return self._leaf_table

    def check_bin_feasibility(self, partial_fixed_bins: Optional[List] = None) -> bool:
        """Is there a feasible completion of the partial fixation of the first len(partial_fixed_bins) binaries
        (mpmilp_program.py:203-237)?"""
        fix = list(partial_fixed_bins or [])
        nb = len(self.binary_indices)
        table = self.leaf_feasibility()
        if len(fix) > nb or any(v not in (0, 1) for v in fix):
            return False
        prefix = 0
        for v in fix:
            prefix = (prefix << 1) | int(v)
        free = nb - len(fix)
        return bool(table[prefix << free:(prefix + 1) << free].any())

    def feasible_combinations(self) -> List[List[int]]:
        """All feasible full fixations, in the order the reference's tree walk lists its leaves."""
        nb = len(self.binary_indices)
        leaves = numpy.flatnonzero(self.leaf_feasibility())
        shifts = numpy.arange(nb - 1, -1, -1, dtype=numpy.int64)[None, :]
        return ((leaves[:, None] >> shifts) & 1).tolist()

    # ---- substituted / relaxed continuous programs (mpmilp_program.py:145-185, 239-269) ---------------------------------------
    def _substituted_rows(self, fixed_combination):
        y = numpy.array(fixed_combination).reshape(-1, 1)
        # which rows carry continuous or parametric content does not depend on the fixation: found once per program (the
        # enumeration substitutes every feasible fixation), dropped whenever the rows change
        key = (id(self.A), id(self.F), self.A.shape, tuple(self.equality_indices))
        cache = getattr(self, '_subst_rows', None)
        if cache is None or cache[0] != key:
            A_cont = self.A[:, self.cont_indices]

            def carries_continuous(i: int) -> bool:
                return not (numpy.allclose(A_cont[i], 0 * A_cont[i]) and numpy.allclose(self.F[i], 0 * self.F[i]))

            eq = [i for i in self.equality_indices if carries_continuous(i)]
            ineq = [i for i in range(self.num_constraints()) if i not in self.equality_indices and carries_continuous(i)]
            kept = [*eq, *ineq]
            cache = (key, A_cont[kept], self.A[:, self.binary_indices][kept], self.b[kept], self.F[kept], len(eq), (self.A, self.F))
            self._subst_rows = cache
        _, A_cont_k, A_bin_k, b_k, F_k, n_eq, _ = cache
        return A_cont_k.copy(), b_k - A_bin_k @ y, F_k.copy(), list(range(n_eq)), y

    def generate_substituted_problem(self, fixed_combination: List[int], deferred: bool = False):
        """The continuous mpLP with the binaries fixed; rows without continuous or parametric content are dropped.  ``deferred``: see
        MPMIQP_Program.generate_substituted_problem."""
        A_cont, b, F, eq, y = self._substituted_rows(fixed_combination)
        c = self.c[self.cont_indices]
        c_c = self.c_c + self.c[self.binary_indices].T @ y
        H_c = self.H[self.cont_indices]
        H_d = self.H[self.binary_indices]
        c_t = self.c_t + (y.T @ H_d).T
        return MPLP_Program(A_cont, b, c, H_c, self.A_t, self.b_t, F, c_c, c_t, self.Q_t, eq, self.solver,
                            post_process=not deferred, _diagnostics=not deferred)

    def _relaxation_rows(self):
        nb = len(self.binary_indices)
        up = numpy.zeros((nb, self.num_x()))
        up[numpy.arange(nb), self.binary_indices] = 1.0
        A = numpy.block([[self.A], [up], [-up]])
        b = numpy.block([[self.b], [numpy.ones((nb, 1))], [numpy.zeros((nb, 1))]])
        F = numpy.block([[self.F], [numpy.zeros((2 * nb, self.num_t()))]])
        return A, b, F

    def generate_relaxed_problem(self, process: bool = True) -> MPLP_Program:
        """Binaries relaxed to [0, 1]."""
        A, b, F = self._relaxation_rows()
        return MPLP_Program(A, b, self.c, self.H, self.A_t, self.b_t, F, self.c_c, self.c_t, self.Q_t,
                            self.equality_indices, self.solver, post_process=process)

    def solve_theta(self, theta_point: numpy.ndarray) -> Optional[SolverOutput]:
        """The MILP at a fixed theta (mpmilp_program.py:187-201)."""
        soln = self.solver.solve_milp(self.c + self.H @ theta_point, self.A, self.b + self.F @ theta_point,
                                      self.equality_indices, self.binary_indices)
        if soln is not None:
            const = self.c_c + self.c_t.T @ theta_point + 0.5 * theta_point.T @ self.Q_t @ theta_point
            soln.obj += float(const[0, 0])
        return soln

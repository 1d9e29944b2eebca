"""Deterministic-solver plug of the host layer.

Mirrors ``Solver`` / ``SolverOutput`` of the reference (solver.py:56-246, solver_interface_utils.py:7-40) for the
problem classes the combinatorial path and its mixed-integer caller need: linear programs, and mixed-integer
linear programs with binary variables.  The only product backend is ``'hip'``: LPs are solved on the MI355X by the
one-wavefront LDS simplex behind ``mpc_lp_solve_batch`` (include/mpcombi.h); a MILP is the batch of LPs over all
fixations of its binaries, one launch (``solve_milp``).  There is no CPU fallback -- without the HIP library or a
GPU, ``solve_lp`` raises ``MpcError``.

``solve_lp`` returns ``None`` unless the LP has an optimal solution, exactly like the reference
(cvxopt_interface.py:20-23,186-198).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy

from . import _lib


@dataclass
class SolverOutput:
    """obj / sol / slack / active_set / dual of an LP solution (solver_interface_utils.py:7-40)."""
    obj: float
    sol: numpy.ndarray
    slack: Optional[numpy.ndarray] = None
    active_set: Optional[numpy.ndarray] = None
    dual: Optional[numpy.ndarray] = None


def _default_solvers() -> Dict[str, str]:
    return {'lp': 'hip', 'milp': 'hip'}


@dataclass
class Solver:
    """Chooses the backend per problem class (LP and binary MILP; both run as LP batches on the device)."""
    solvers: Dict[str, str] = field(default_factory=_default_solvers)
    device: int = 0

    supported_problems = ('lp', 'milp')
    supported_solvers = ('hip',)

    def __post_init__(self):
        for problem, backend in self.solvers.items():
            if problem not in self.supported_problems:
                raise RuntimeError(f'Problem {problem} is not supported! ppopt_amd supports {self.supported_problems}')
            if backend not in self.supported_solvers:
                raise RuntimeError(f'Solver {backend} is not supported! ppopt_amd supports {self.supported_solvers}')

    def solve_lp(self, c: Optional[numpy.ndarray], A: Optional[numpy.ndarray], b: Optional[numpy.ndarray],
                 equality_constraints: Optional[Sequence[int]] = None, verbose: bool = False,
                 get_duals: bool = True) -> Optional[SolverOutput]:
        """min c'x s.t. Ax <= b, rows ``equality_constraints`` as equalities, x free."""
        if A is None or A.shape[0] == 0 or A.shape[1] == 0:
            return None
        out = self.solve_lp_batch(c, A, b, [list(equality_constraints or [])])
        return out[0]

    def solve_lp_batch(self, c, A, b, equality_sets: List[Sequence[int]]) -> List[Optional[SolverOutput]]:
        """The same LP data with a different equality set per instance (what presolve needs,
        constraint_utilities.py:186-200): one device launch, one wavefront per instance."""
        A = numpy.ascontiguousarray(A, dtype=numpy.float64)
        m, n = A.shape
        bb = numpy.ascontiguousarray(b, dtype=numpy.float64).reshape(-1)
        flags = numpy.zeros((len(equality_sets), m), dtype=numpy.uint8)
        for i, eq in enumerate(equality_sets):
            flags[i, list(eq)] = 1
        cc = None if c is None else numpy.ascontiguousarray(c, dtype=numpy.float64).reshape(-1)
        status, x, obj, _ = _lib.lp_solve_batch(A, bb, cc, flags, device=self.device)
        res: List[Optional[SolverOutput]] = []
        for i in range(len(equality_sets)):
            if status[i] != _lib.LP_OPTIMAL:
                res.append(None)
                continue
            slack = bb - A @ x[i]
            active = numpy.nonzero(numpy.abs(slack) <= 1e-10)[0]
            res.append(SolverOutput(float(obj[i]), x[i].copy(), slack, active, None))
        return res

    # ---- binary MILPs as LP batches (solver.py:248-282; the reference hands these to Gurobi) -------------------------
    MAX_BINARIES = 20

    @staticmethod
    def binary_fixations(n_bin: int) -> numpy.ndarray:
        """All 2^n_bin fixations, [2^n_bin, n_bin], first binary most significant: row order == the order in which
        MITree.get_full_leafs lists leaves (0-branch before 1-branch at every depth, mitree.py:84-102)."""
        idx = numpy.arange(1 << n_bin, dtype=numpy.int64)[:, None]
        shifts = numpy.arange(n_bin - 1, -1, -1, dtype=numpy.int64)[None, :]
        return ((idx >> shifts) & 1).astype(numpy.int32)

    def solve_milp_batch(self, c, A, b, equality_sets: List[Sequence[int]], bin_vars: Sequence[int]):
        """For every equality set, the LP of every binary fixation: returns (status, x, obj) shaped
        [n_sets, 2^n_bin(, n)] -- status == 0 where that fixation's LP has an optimal solution.

        The fixation is expressed through equality flags on appended rows  y_j <= 1  and  -y_j <= 0  (flagging the
        first fixes y_j = 1, the second y_j = 0), so that all instances share one matrix and one right-hand side."""
        A = numpy.ascontiguousarray(A, dtype=numpy.float64)
        m, n = A.shape
        bins = list(bin_vars or [])
        nb = len(bins)
        if nb > self.MAX_BINARIES:
            raise ValueError(f'{nb} binary variables: the enumeration backend is limited to {self.MAX_BINARIES}')
        bb = numpy.ascontiguousarray(b, dtype=numpy.float64).reshape(-1)
        up = numpy.zeros((nb, n))
        up[numpy.arange(nb), bins] = 1.0
        A_aug = numpy.vstack([A, up, -up])
        b_aug = numpy.concatenate([bb, numpy.ones(nb), numpy.zeros(nb)])
        fix = self.binary_fixations(nb)                                   # [2^nb, nb]
        n_fix, n_sets = fix.shape[0], len(equality_sets)
        flags = numpy.zeros((n_sets, n_fix, m + 2 * nb), dtype=numpy.uint8)
        for i, eq in enumerate(equality_sets):
            flags[i, :, list(eq)] = 1
        flags[:, :, m:m + nb] = (fix == 1)[None]
        flags[:, :, m + nb:] = (fix == 0)[None]
        cc = None if c is None else numpy.ascontiguousarray(c, dtype=numpy.float64).reshape(-1)
        status, x, obj, _ = _lib.lp_solve_batch(A_aug, b_aug, cc, flags.reshape(n_sets * n_fix, -1),
                                                device=self.device)
        x = x.reshape(n_sets, n_fix, n)
        x[:, :, bins] = fix[None]
        return status.reshape(n_sets, n_fix), x, obj.reshape(n_sets, n_fix)

    def solve_milp(self, c: Optional[numpy.ndarray], A: Optional[numpy.ndarray], b: Optional[numpy.ndarray],
                   equality_constraints: Optional[Sequence[int]] = None, bin_vars: Optional[Sequence[int]] = None,
                   verbose: bool = False, get_duals: bool = True) -> Optional[SolverOutput]:
        """min c'[x,y] s.t. A[x,y] <= b, rows ``equality_constraints`` as equalities, y binary, x free.  ``None``
        unless some fixation has an optimal LP; the best objective wins, the first fixation on ties."""
        if A is None or A.shape[0] == 0 or A.shape[1] == 0:
            return None
        status, x, obj = self.solve_milp_batch(c, A, b, [list(equality_constraints or [])], bin_vars)
        ok = status[0] == _lib.LP_OPTIMAL
        if not ok.any():
            return None
        best = int(numpy.argmin(numpy.where(ok, obj[0], numpy.inf)))
        sol = x[0, best].copy()
        bb = numpy.asarray(b, dtype=numpy.float64).reshape(-1)
        slack = bb - numpy.asarray(A, dtype=numpy.float64) @ sol
        return SolverOutput(float(obj[0, best]), sol, slack, numpy.nonzero(numpy.abs(slack) <= 1e-10)[0], None)

"""Deterministic-solver plug of the host layer.

Mirrors ``Solver`` / ``SolverOutput`` of the reference (solver.py:56-246, solver_interface_utils.py:7-40) for the
one problem class the combinatorial path needs: linear programs.  The only product backend is ``'hip'``: the LP is
solved on the MI355X by the one-wavefront LDS simplex behind ``mpc_lp_solve_batch`` (include/mpcombi.h).  There is
no CPU fallback -- without the HIP library or a GPU, ``solve_lp`` raises ``MpcError``.

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
    return {'lp': 'hip'}


@dataclass
class Solver:
    """Chooses the backend per problem class; only LPs are on the combinatorial path."""
    solvers: Dict[str, str] = field(default_factory=_default_solvers)
    device: int = 0

    supported_problems = ('lp',)
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

"""CriticalRegion: the output record of the combinatorial path.

Field names, shapes and index conventions are those of the reference dataclass (critical_region.py:9-48):
x*(theta) = A theta + b,  lambda*(theta) = C theta + d (one row per active constraint, equalities first),
region {theta: E theta <= f} with unit-norm rows; ``omega_set`` indexes rows of A_t, ``lambda_set`` holds
constraint ids whose multiplier bounds the region, ``regular_set = [[index into inactive list], [constraint ids]]``.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy


@dataclass(eq=False)
class CriticalRegion:
    A: numpy.ndarray
    b: numpy.ndarray
    C: numpy.ndarray
    d: numpy.ndarray
    E: numpy.ndarray
    f: numpy.ndarray
    active_set: List[int]

    omega_set: List[int] = field(default_factory=list)
    lambda_set: List[int] = field(default_factory=list)
    regular_set: List[List[int]] = field(default_factory=list)

    y_fixation: Optional[numpy.ndarray] = None
    y_indices: Optional[numpy.ndarray] = None
    x_indices: Optional[numpy.ndarray] = None

    def __repr__(self):
        sets = (f'active set {self.active_set}; bounded by parameter rows {self.omega_set}, multipliers of {self.lambda_set}, '
                f'inactive constraints {self.regular_set}')
        blocks = '\n'.join(f'{name} =\n{numpy.asarray(value)}' for name, value in
                           (('A', self.A), ('b', self.b), ('C', self.C), ('d', self.d), ('E', self.E), ('f', self.f)))
        return f'CriticalRegion: x = A theta + b, lambda = C theta + d on E theta <= f\n{sets}\n{blocks}'

    def evaluate(self, theta: numpy.ndarray) -> numpy.ndarray:
        """x*(theta); binaries of a mixed-integer parent are spliced in when present (critical_region.py:64-77)."""
        x = self.A @ theta + self.b
        if self.y_fixation is None:
            return x
        full = numpy.zeros((len(self.x_indices) + len(self.y_indices),))
        full[self.x_indices] = x.flatten()
        full[self.y_indices] = self.y_fixation
        return full.reshape(-1, 1)

    def lagrange_multipliers(self, theta: numpy.ndarray) -> numpy.ndarray:
        return self.C @ theta + self.d

    def is_inside(self, theta: numpy.ndarray, tol: float = 1e-5) -> bool:
        return bool(numpy.all(self.E @ theta - self.f < tol))

    def is_full_dimension(self, solver=None) -> bool:
        """Chebyshev radius of {E theta <= f} above 1e-8 (critical_region.py:89-105)."""
        from .utils.chebyshev_ball import chebyshev_ball
        sol = chebyshev_ball(self.E, self.f, solver=solver)
        return sol is not None and bool(sol.sol[-1] > 10 ** -8)

    def get_constraints(self):
        return [self.E, self.f]

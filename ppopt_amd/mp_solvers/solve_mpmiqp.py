"""Entry point for mixed-integer programs (the role of the reference's mp_solvers/solve_mpmiqp.py:14-66).

``solve_mpmiqp(problem)`` runs the chosen mixed-integer strategy -- enumeration of the binary fixations is the only
one, as in the reference -- with ``cont_algo`` for the continuous sub-programs, then removes the overlaps between
fixations where that is defined: one parameter, linear objective, no bilinear theta-x terms.
"""
from enum import Enum

import numpy

from ..solution import Solution
from ..utils.region_overlap_utils import reduce_overlapping_critical_regions_1d
from . import mpmiqp_enumeration
from .solve_mpqp import mpqp_algorithm, solve_mpqp


class mpmiqp_algorithm(Enum):
    """Strategies for the integer part; pass one as ``mpmiqp_algo``."""
    enumerate = 'enumerate'

    def __str__(self):
        return self.name

    @staticmethod
    def all_algos():
        return '\n'.join(f'mpmiqp_algorithm.{member}' for member in mpmiqp_algorithm) + '\n'


_STRATEGIES = {mpmiqp_algorithm.enumerate: mpmiqp_enumeration.solve_mpmiqp_enumeration}


def _overlaps_can_be_reduced(problem) -> bool:
    """The interval arithmetic of region_overlap_utils needs one parameter and objectives that are affine in theta."""
    if problem.num_t() != 1 or hasattr(problem, 'Q'):
        return False
    bilinear_mass = numpy.sum(numpy.abs(problem.H[problem.cont_indices, :]))
    return bool(numpy.isclose(bilinear_mass, 0))


def solve_mpmiqp(problem, mpmiqp_algo: mpmiqp_algorithm = mpmiqp_algorithm.enumerate,
                 cont_algo: mpqp_algorithm = mpqp_algorithm.combinatorial, num_cores=-1,
                 reduce_overlap=True) -> Solution:
    if not problem.binary_indices:
        print('No binary variables in this program: it is solved as a continuous one.')
        return solve_mpqp(problem, cont_algo)
    if mpmiqp_algo not in _STRATEGIES:
        raise TypeError('mpmiqp_algo has to be a member of ppopt_amd.mp_solvers.solve_mpmiqp.mpmiqp_algorithm; '
                        f'available:\n{mpmiqp_algorithm.all_algos()}')
    solution = _STRATEGIES[mpmiqp_algo](problem, num_cores, cont_algo)
    if reduce_overlap and _overlaps_can_be_reduced(problem):
        regions, still_overlapping = reduce_overlapping_critical_regions_1d(problem, solution.critical_regions)
        return Solution(problem, regions, still_overlapping)
    return solution

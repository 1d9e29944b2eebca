"""Algorithm dispatch for mixed-integer programs: ``solve_mpmiqp`` (reference: mp_solvers/solve_mpmiqp.py:14-66)."""
from enum import Enum

import numpy

from ..solution import Solution
from ..utils.region_overlap_utils import reduce_overlapping_critical_regions_1d
from .mpmiqp_enumeration import solve_mpmiqp_enumeration
from .solve_mpqp import mpqp_algorithm, solve_mpqp


class mpmiqp_algorithm(Enum):
    enumerate = 'enumerate'

    def __str__(self):
        return self.name

    @staticmethod
    def all_algos():
        return ''.join(f'mpmiqp_algorithm.{a}\n' for a in mpmiqp_algorithm)


def solve_mpmiqp(problem, mpmiqp_algo: mpmiqp_algorithm = mpmiqp_algorithm.enumerate,
                 cont_algo: mpqp_algorithm = mpqp_algorithm.combinatorial, num_cores=-1,
                 reduce_overlap=True) -> Solution:
    if len(problem.binary_indices) == 0:
        print('The problem does not have any binary variables, solving as a continuous problem instead.')
        return solve_mpqp(problem, cont_algo)
    if not isinstance(mpmiqp_algo, mpmiqp_algorithm):
        raise TypeError('You must pass an algorithm from mpmiqp_algorithm as the continuous algorithm. These can be '
                        'found by importing the following \n\nfrom ppopt_amd.mp_solvers.solve_mpmiqp import '
                        f'mpmiqp_algorithm\n\nWith the following choices\n{mpmiqp_algorithm.all_algos()}')

    cand_sol = Solution(problem, [])
    if mpmiqp_algo == mpmiqp_algorithm.enumerate:
        cand_sol = solve_mpmiqp_enumeration(problem, num_cores, cont_algo)

    # overlaps are only resolved for 1-D mpMILPs without bilinear terms (solve_mpmiqp.py:55-64)
    bilinear = not numpy.isclose(numpy.sum(numpy.abs(problem.H[problem.cont_indices, :])), 0)
    if not (problem.num_t() > 1 or hasattr(problem, 'Q') or not reduce_overlap or bilinear):
        regions, still_overlapping = reduce_overlapping_critical_regions_1d(problem, cand_sol.critical_regions)
        return Solution(problem, regions, still_overlapping)
    return cand_sol

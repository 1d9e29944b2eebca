"""Mixed-integer enumeration: the caller of the combinatorial path that SURVEY.md §8(f) ranks second
(reference: mp_solvers/mpmiqp_enumeration.py:12-64).

  1. the feasible binary fixations -- one device batch of LPs (MPMILP_Program.leaf_feasibility);
  2. per fixation the substituted continuous mpLP/mpQP, presolved (its redundancy LPs are a device batch) and solved
     by the MI355X combinatorial path (`solve_mpqp`);
  3. regions tagged with their fixation and concatenated, in fixation order.

``num_cores`` is accepted for signature compatibility: the sub-problems run one after the other on the device
(each is itself a batch over candidates).
"""
from ..solution import Solution
from .solve_mpqp import mpqp_algorithm, solve_mpqp


def solve_mpmiqp_enumeration(program, num_cores: int = -1,
                             cont_algorithm: mpqp_algorithm = mpqp_algorithm.combinatorial) -> Solution:
    feasible_combinations = program.feasible_combinations()
    sols = [solve_mpqp(program.generate_substituted_problem(fix), cont_algorithm) for fix in feasible_combinations]

    collected = []
    for fix, sol in zip(feasible_combinations, sols):
        for region in sol.critical_regions:
            region.y_fixation = fix
            region.y_indices = program.binary_indices
            region.x_indices = program.cont_indices
        collected.extend(sol.critical_regions)
    return Solution(program, collected, is_overlapping=True)

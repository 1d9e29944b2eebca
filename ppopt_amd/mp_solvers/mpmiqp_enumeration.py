"""Mixed-integer enumeration: the caller of the combinatorial path that SURVEY.md §8(f) ranks second
(reference: mp_solvers/mpmiqp_enumeration.py:12-64).

  1. the feasible binary fixations -- one device batch of LPs (MPMILP_Program.leaf_feasibility);
  2. per fixation the substituted continuous mpLP/mpQP, presolved (its redundancy LPs are a device batch) and solved
     by the MI355X combinatorial path (`solve_mpqp`);
  3. regions tagged with their fixation and concatenated, in fixation order.

``num_cores`` keeps its meaning of "sub-problems in flight": the reference maps them over a process pool
(mpmiqp_enumeration.py:46-50); here each in-flight sub-problem is a host thread driving its own device handle and
HIP streams (the C ABI releases the GIL, handles are independent, pools are locked), so the small kernels of
different sub-problems overlap on the GPU and one sub-problem's host bookkeeping hides behind another's kernels.
``num_cores=1`` runs them one after the other; -1 picks min(8, host cores).
"""
import warnings
from concurrent.futures import ThreadPoolExecutor

from ..solution import Solution
from ..utils.general_utils import num_cpu_cores
from .solve_mpqp import mpqp_algorithm, solve_mpqp


def solve_mpmiqp_enumeration(program, num_cores: int = -1,
                             cont_algorithm: mpqp_algorithm = mpqp_algorithm.combinatorial) -> Solution:
    feasible_combinations = program.feasible_combinations()
    if num_cores == -1:
        num_cores = min(8, num_cpu_cores())

    def solve_fixation(fix):
        with warnings.catch_warnings():        # the substituted program repeats the parent's construction warnings
            warnings.simplefilter('ignore')
            sub = program.generate_substituted_problem(fix)
            try:
                return solve_mpqp(sub, cont_algorithm)
            finally:
                sub.release_engine()       # the handle's device blocks go straight to the next sub-problem

    if num_cores <= 1 or len(feasible_combinations) <= 1:
        sols = [solve_fixation(fix) for fix in feasible_combinations]
    else:
        with ThreadPoolExecutor(max_workers=num_cores) as pool:
            sols = list(pool.map(solve_fixation, feasible_combinations))       # results in fixation order

    collected = []
    for fix, sol in zip(feasible_combinations, sols):
        for region in sol.critical_regions:
            region.y_fixation = fix
            region.y_indices = program.binary_indices
            region.x_indices = program.cont_indices
        collected.extend(sol.critical_regions)
    return Solution(program, collected, is_overlapping=True)

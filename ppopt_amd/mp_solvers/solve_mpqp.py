"""Algorithm dispatch: ``solve_mpqp(problem, algorithm)`` (reference: mp_solvers/solve_mpqp.py:23-114).

The enum keeps every member of the reference's ``mpqp_algorithm`` so that user code keeps importing and selecting
algorithms the same way.  The three combinatorial members run on the MI355X (they differ in the reference only in
how the CPU work is scheduled and pruned; all give the same region set), and so does
``combinatorial_graph`` and the four ``graph`` members (the connected-graph traversals of mpqp_combi_graph.py and
mpqp_graph.py on the same device kernels, mpqp_hip_combi_graph.py) and the three ``geometric`` members (facet centres as
an LP batch, probes as a QP batch, mpqp_hip_geometric.py); the graph and geometric drivers cover mpQPs.
"""
from enum import Enum

import numpy

from ..mplp_program import MPLP_Program
from ..mpqp_program import MPQP_Program
from ..solution import Solution
from . import mpqp_hip_combi_graph, mpqp_hip_combinatorial, mpqp_hip_geometric


class mpqp_algorithm(Enum):
    combinatorial = 'combinatorial'
    combinatorial_parallel = 'p combinatorial'
    combinatorial_parallel_exp = 'p combinatorial exp'
    graph = 'graph'
    graph_exp = 'graph exp'
    graph_parallel = 'p graph'
    graph_parallel_exp = 'p graph exp'
    combinatorial_graph = 'combinatorial graph'
    geometric = 'geometric'
    geometric_parallel = 'p geometric'
    geometric_parallel_exp = 'p geometric exp'

    def __str__(self):
        return self.name

    @staticmethod
    def all_algos():
        return ''.join(f'mpqp_algorithm.{a}\n' for a in mpqp_algorithm)


_COMBINATORIAL = (mpqp_algorithm.combinatorial, mpqp_algorithm.combinatorial_parallel,
                  mpqp_algorithm.combinatorial_parallel_exp)
_GEOMETRIC = (mpqp_algorithm.geometric, mpqp_algorithm.geometric_parallel, mpqp_algorithm.geometric_parallel_exp)
_GRAPH = (mpqp_algorithm.graph, mpqp_algorithm.graph_exp, mpqp_algorithm.graph_parallel, mpqp_algorithm.graph_parallel_exp)


def solve_mpqp(problem: MPQP_Program, algorithm: mpqp_algorithm = mpqp_algorithm.combinatorial) -> Solution:
    if not isinstance(algorithm, mpqp_algorithm):
        raise TypeError('You must pass an algorithm from mpqp_algorithm as the continuous algorithm. These can be '
                        'found by importing the following \n\nfrom ppopt_amd.mp_solvers.solve_mpqp import '
                        f'mpqp_algorithm\n\nWith the following choices\n{mpqp_algorithm.all_algos()}')
    # the device the program's presolve LPs ran on (Solver(device=...)) is the device the solve runs on
    device = int(getattr(getattr(problem, 'solver', None), 'device', 0) or 0)
    if algorithm is mpqp_algorithm.combinatorial_graph:
        solution = mpqp_hip_combi_graph.solve(problem, device=device)
    elif algorithm in _GRAPH:
        solution = mpqp_hip_combi_graph.solve_graph(problem, device=device)
    elif algorithm in _GEOMETRIC:
        solution = mpqp_hip_geometric.solve(problem, device=device)
    else:
        # the serial driver (and the _exp parallel one) expand every feasible set; the parallel driver also prunes the supersets of
        # sets that are optimal with a lower-dimensional region (driver differences)
        solution = mpqp_hip_combinatorial.solve(problem, device=device, prune_lowdim=algorithm is mpqp_algorithm.combinatorial_parallel)
    # overlap flags exactly as the reference sets them (solve_mpqp.py:103-112)
    if isinstance(problem, MPQP_Program) and min(numpy.linalg.eigvalsh(problem.Q)) <= 0:
        solution.is_overlapping = True
    if isinstance(problem, MPLP_Program):
        solution.is_overlapping = True
    return solution

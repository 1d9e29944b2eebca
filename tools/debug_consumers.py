import sys, warnings, traceback
sys.path.insert(0, '.')
import numpy
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combi_graph as G
from ppopt_amd.solution import Solution
for args in ((5, 1, 17, 225226), (7, 2, 20, 985440)):
    d = pg.generate_mpqp_data(*args)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
    gsol = G.solve_graph(prog)
    nt = prog.num_t()
    pts = (-1 + 2 * numpy.random.default_rng(0).random((500, nt))) * 1.05 * numpy.abs(prog.b_t).max()
    Solution.WALK_MIN_REGIONS = 1
    try:
        x, i = gsol.evaluate_batch(pts)
        print(args, 'located', (i >= 0).sum(), 'adjacency', gsol.locator().has_adjacency)
        inside = numpy.flatnonzero(i >= 0)[:50]
        res = prog.solve_theta_batch(pts[inside])
        print('qp ok', sum(r is not None for r in res))
    except Exception:
        traceback.print_exc()

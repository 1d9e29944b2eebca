"""Do all drivers return the same complete solution?  Random small mpQPs (run on the GPU box):
    python tools/algo_agreement.py [n_programs] [seed]
combinatorial (complete), graph, combinatorial_graph and geometric are compared as sets of active sets."""
import sys, warnings
sys.path.insert(0, '.')
import numpy
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as C, mpqp_hip_combi_graph as G, mpqp_hip_geometric as GE
n_prob = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = numpy.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
tot = 0
n_walk = 0
for it in range(n_prob):
    nx, nt = int(rng.integers(2, 8)), int(rng.integers(1, 5))
    m = int(rng.integers(nx + 2, 3 * nx + 4))
    seed = int(rng.integers(0, 10 ** 6))
    d = pg.generate_mpqp_data(nx, nt, m, seed)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
    ref = {tuple(r.active_set) for r in C.solve(prog, prune_lowdim=False).critical_regions}     # the serial driver's rule
    par = {tuple(r.active_set) for r in C.solve(prog).critical_regions}                           # the parallel driver's rule
    res = {}
    for label, fn in (('graph', G.solve_graph), ('combinatorial_graph', G.solve), ('geometric', GE.solve)):
        try:
            res[label] = {tuple(r.active_set) for r in fn(prog).critical_regions}
        except Exception as e:
            res[label] = e
    line = f'({nx},{nt},{m},{seed}) combinatorial serial rule {len(ref)}, parallel rule {len(par)}'
    if par != ref:
        print('NOTE parallel pruning rule loses regions:', line, flush=True)
    ok = True
    for label, k in res.items():
        if isinstance(k, Exception):
            line += f' | {label}: {type(k).__name__} {k}'
            ok = False
        else:
            line += f' | {label} {len(k)} (-{len(ref - k)} +{len(k - ref)})'
            ok = ok and k == ref
    # consumers on the complete solution: the walk locator against the list scan, the QP at points against the explicit law
    try:
        from ppopt_amd.solution import Solution
        gsol = G.solve_graph(prog)
        half = 1.05 * numpy.abs(d['b_t']).max()          # the generator's parameter box (presolve may have dropped redundant rows of it)
        lo, hi = -half * numpy.ones(nt), half * numpy.ones(nt)
        pts = lo + rng.random((1500, nt)) * (hi - lo)
        Solution.WALK_MIN_REGIONS = 1
        gsol.use_walk = True
        xw, iw = gsol.evaluate_batch(pts)
        walked = gsol.locator().has_adjacency
        gsol.use_walk = False
        xs, i_s = gsol.evaluate_batch(pts)
        if not numpy.array_equal(iw, i_s):
            ok = False
            line += f' | WALK differs from scan on {(iw != i_s).sum()} of {len(pts)} points'
        inside = numpy.flatnonzero(i_s >= 0)[:300]
        res = prog.solve_theta_batch(pts[inside])
        worst = 0.0
        for p, r in zip(inside, res):
            if r is None:
                worst = numpy.inf
            else:
                worst = max(worst, float(numpy.max(numpy.abs(r.sol - xs[p]) / (1 + numpy.abs(xs[p])))))
        if worst > 1e-7:
            ok = False
            line += f' | QP differs from the explicit law by {worst:.2e}'
        n_walk += int(walked)
    except Exception as e:
        ok = False
        line += f' | consumers: {type(e).__name__} {e}'
    tot += 1
    if not ok:
        bad += 1
        print('DIFF', line, flush=True)
    prog.release_engine()
print(f'{tot} programs, {bad} with differences; walk locator exercised on {n_walk}')

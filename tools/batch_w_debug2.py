"""Debug: regions of solve_many against solve whose facet lists differ, against the CPU oracle's lists at three LP tolerances (GPU box)."""
import os, sys, warnings
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
from oracle import oracle as orc
warnings.simplefilter('ignore')
seeds = [int(v) for v in sys.argv[1:]] or [21, 22, 23, 24, 25, 26, 27, 28]
def programs():
    out = []
    for seed in seeds:
        d = pg.generate_mpqp_data(8, 4, 16, seed)
        out.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F']))
    return out
one = [mpqp_hip_combinatorial.solve(p) for p in programs()]
progs = programs()
many = mpqp_hip_combinatorial.solve_many(progs)
sets = lambda r: (tuple(r.omega_set), tuple(r.lambda_set), tuple(map(tuple, r.regular_set)))
osets = lambda r: (tuple(r['omega_set']), tuple(r['lambda_set']), tuple(map(tuple, r['regular_set'])))
tot = [0, 0, 0, 0]
for n, (a, b, p) in enumerate(zip(one, many, progs)):
    P = orc.OracleProblem(p.A, p.b, p.F, p.c, p.H, p.Q, p.A_t, p.b_t, len(p.equality_indices))
    ka = {tuple(r.active_set): r for r in a.critical_regions}; kb = {tuple(r.active_set): r for r in b.critical_regions}
    assert ka.keys() == kb.keys()
    for key, r1 in ka.items():
        r2 = kb[key]
        tot[0] += 1
        if sets(r1) == sets(r2):
            continue
        res = {}
        for tol in (1e-9, 1e-7, 1e-5):
            orc.set_feas_tol(tol)
            v, reg = P.gen_cr_from_active_set(list(key))
            res[tol] = osets(reg) if reg else None
        orc.set_feas_tol(1e-7)
        flips = len({res[t] for t in res}) > 1
        tot[1] += 1; tot[2] += sets(r1) == res[1e-7]; tot[3] += sets(r2) == res[1e-7]
        M = numpy.block([[p.A[list(key)], numpy.zeros((len(key), len(key)))], [p.Q, p.A[list(key)].T]])
        print('program', n, 'active', key, 'cond %.1e' % numpy.linalg.cond(M), 'oracle flips with tol', flips, '| single == oracle(1e-7)', sets(r1) == res[1e-7], '| batch == oracle(1e-7)', sets(r2) == res[1e-7],
              '| rows single/batch/oracle', len(r1.E), len(r2.E), [sum(len(x) if not isinstance(x[0] if x else 0, tuple) else len(x[0]) for x in res[t]) if res[t] else None for t in res])
print('regions', tot[0], 'differing', tot[1], 'single agrees with oracle', tot[2], 'batch agrees', tot[3])

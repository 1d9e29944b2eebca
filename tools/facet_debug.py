import sys, warnings
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy
from scipy.optimize import linprog
from ppopt_amd import MPQP_Program, Solver, problem_generator as pg
from oracle import oracle as orc
orc.build()
nx, nt, m, seed, ne = (int(v) for v in sys.argv[1:6])
key = [int(v) for v in sys.argv[6:]]
d = pg.generate_mpqp_data(nx, nt, m, seed)
if ne: d['equality_indices'] = list(range(ne))
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=d.get('equality_indices'), solver=Solver())
P = orc.OracleProblem(prog.A, prog.b, prog.F, prog.c, prog.H, prog.Q, prog.A_t, prog.b_t, len(prog.equality_indices))
ost, oregs = P.check_level(numpy.array([key], dtype=numpy.int32), 0, True)
q = oregs[0]
eng = prog.engine(0)
status, rd, ri, _, _ = eng.check_level(numpy.array([key], dtype=numpy.int32), numpy.zeros((0, 2), dtype=numpy.uint64), False)
from ppopt_amd.mp_solvers.mpqp_hip_combinatorial import unpack_regions
r = unpack_regions(rd, ri, eng.n_x, eng.n_t, eng.n_c, eng.n_tc)[0]
print('oracle omega', q['omega_set'], 'lambda', q['lambda_set'], 'regular', q['regular_set'])
print('gpu    omega', r.omega_set, 'lambda', r.lambda_set, 'regular', r.regular_set)
print('E rows oracle', q['E'].shape[0], 'gpu', r.E.shape[0]); sys.exit(0)
# ground truth: all rows of the region polytope built with numpy, min slack of each row over the polytope
A, b, F, c, H, Q, At, bt = prog.A, prog.b, prog.F, prog.c, prog.H, prog.Q, prog.A_t, prog.b_t
k = len(key)
K = numpy.block([[Q, A[key].T], [A[key], numpy.zeros((k, k))]])
sol = numpy.linalg.solve(K, numpy.block([[-c, -H], [b[key], F[key]]]))
X, L = sol[:nx], sol[nx:]
inact = [i for i in range(A.shape[0]) if i not in key]
rows = [(-L[i, 1:], L[i, 0], ('lam', key[i])) for i in range(k)] + [(A[i] @ X[:, 1:] - F[i], b[i, 0] - A[i] @ X[:, 0], ('reg', i)) for i in inact] + [(At[i], bt[i, 0], ('om', i)) for i in range(At.shape[0])]
E = numpy.array([g for g, h, _ in rows]); f = numpy.array([h for g, h, _ in rows])
nrm = numpy.linalg.norm(E, axis=1); keep = nrm > 1e-8
E, f, tags = E[keep] / nrm[keep, None], f[keep] / nrm[keep], [t for (g, h, t), kk in zip(rows, keep) if kk]
for i, t in enumerate(tags):
    res = linprog(E[i], A_ub=E, b_ub=f, bounds=(None, None), method='highs')   # max E_i theta  <=> min slack
    slack = f[i] + res.fun * -1 if res.status == 0 else None
    ms = f[i] - (-res.fun) if res.status == 0 else None
    print(t, 'min slack %.3e' % ms if ms is not None else res.message)

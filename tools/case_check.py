"""One fuzz case under several switches: python tools/case_check.py nx nt m seed "a,b;a,b,c" [mpqp|open-mode ...]
prints the GPU status of the listed candidates (each in its own process per environment), the oracle's verdict and the margin of the
(x,theta) feasibility question (largest t with: inactive rows' slack >= t, A_t theta <= b_t - t, active rows equal) by scipy's HiGHS."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, warnings
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import numpy
from ppopt_amd import MPQP_Program, Solver, problem_generator as pg
nx, nt, m, seed = %d, %d, %d, %d
want = %r
d = pg.generate_mpqp_data(nx, nt, m, seed)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], solver=Solver())
eng = prog.engine(0)
eng.pruned_clear(); eng.frontier_root()
out = {}
depth = max(len(w) for w in want)
for lev in range(depth):
    st = eng.level_run(True)
    gc, gs = eng.frontier_get(), eng.level_status()
    idx = {tuple(r): i for i, r in enumerate(gc.tolist())}
    for w in want:
        if len(w) == lev + 1:
            out[str(list(w))] = int(gs[idx[tuple(w)]]) if tuple(w) in idx else None
    eng.frontier_advance()
print(json.dumps(out))
'''
def run(env, args, want):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, '-c', CHILD % ((ROOT, ROOT) + tuple(args) + (want,))], env=e, capture_output=True, text=True)
    if r.returncode: print(r.stderr[-1500:]); raise SystemExit(1)
    return json.loads(r.stdout.strip().splitlines()[-1])
if __name__ == '__main__':
    args = [int(v) for v in sys.argv[1:5]]
    want = [tuple(int(x) for x in s.split(',')) for s in sys.argv[5].split(';')]
    for name, env in (('default', {}), ('round-4 paths', {'MPC_XQ_THREAD': '0', 'MPC_X1': '0', 'MPC_NO_KKT_LISTS': '1', 'MPC_NO_SMALL_FUSE': '1', 'MPC_NO_X_FIRST': '1'}),
                      ('classic, no quick test', {'MPC_NO_SMALLPATH': '1', 'MPC_NO_XQUICK': '1', 'MPC_XQ_THREAD': '0', 'MPC_X1': '0'}), ('LDS engine only', {'MPC_FORCE_V1': '1'})):
        print('%-24s' % name, run(env, args, want))
    sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
    import numpy, warnings
    from scipy.optimize import linprog
    from ppopt_amd import MPQP_Program, Solver, problem_generator as pg
    from oracle import oracle as orc
    orc.build()
    d = pg.generate_mpqp_data(*args)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], solver=Solver())
    P = orc.OracleProblem(prog.A, prog.b, prog.F, prog.c, prog.H, prog.Q, prog.A_t, prog.b_t, len(prog.equality_indices))
    A, b, F, At, bt = prog.A, prog.b.ravel(), prog.F, prog.A_t, prog.b_t.ravel()
    nx, nt = A.shape[1], F.shape[1]
    for w in want:
        ost, _ = P.check_level(numpy.array([w], dtype=numpy.int32), 0, False)
        act = list(w); ina = [i for i in range(A.shape[0]) if i not in act]
        # variables x, theta, t ; maximise t
        c = numpy.zeros(nx + nt + 1); c[-1] = -1
        Aub = numpy.vstack([numpy.hstack([A[ina], -F[ina], numpy.ones((len(ina), 1))]), numpy.hstack([numpy.zeros((At.shape[0], nx)), At, numpy.ones((At.shape[0], 1))])])
        bub = numpy.concatenate([b[ina], bt])
        Aeq = numpy.hstack([A[act], -F[act], numpy.zeros((len(act), 1))]); beq = b[act]
        r = linprog(c, A_ub=Aub, b_ub=bub, A_eq=Aeq, b_eq=beq, bounds=[(None, None)] * (nx + nt) + [(None, 1.0)], method='highs')
        print(w, 'oracle', int(ost[0]), 'feasibility margin t* =', (-r.fun if r.status == 0 else r.message))

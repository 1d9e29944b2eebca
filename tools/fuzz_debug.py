import sys, warnings
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy
from ppopt_amd import MPQP_Program, Solver, problem_generator as pg
from oracle import oracle as orc
from conftest import is_knife_edge, kkt_condition
nx, nt, m, seed = (int(v) for v in sys.argv[1:5])
d = pg.generate_mpqp_data(nx, nt, m, seed)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], solver=Solver())
orc.build()
P = orc.OracleProblem(prog.A, prog.b, prog.F, prog.c, prog.H, prog.Q, prog.A_t, prog.b_t, len(prog.equality_indices))
olevels, oregions, _ = P.solve(0, True, None)
eng = prog.engine(0)
eng.pruned_clear(); eng.frontier_root()
for depth, (oc, ost) in enumerate(olevels):
    st = eng.level_run(True)
    gc, gs = eng.frontier_get(), eng.level_status()
    print('level', depth, 'oracle', len(oc), numpy.bincount(ost, minlength=6).tolist(), 'gpu', len(gc), numpy.bincount(gs, minlength=6).tolist())
    ref = {tuple(c): int(v) for c, v in zip(oc.tolist(), ost.tolist())}
    for c, v in zip(gc.tolist(), gs.tolist()):
        if tuple(c) in ref and ref[tuple(c)] != v:
            print('   ', c, 'gpu', v, 'oracle', ref[tuple(c)], 'knife-edge' if is_knife_edge(P, c) else 'ROBUST', 'cond %.2e' % kkt_condition(P, c))
    if st.n_children == 0: break
    eng.frontier_advance()

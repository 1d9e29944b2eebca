import sys, warnings
sys.path.insert(0, '.')
import numpy
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as C, mpqp_hip_combi_graph as G
d = pg.generate_mpqp_data(7, 4, 24, 623692)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
print('n_c', prog.num_constraints(), 'n_eq', len(prog.equality_indices))
g = {tuple(r.active_set) for r in G.solve_graph(prog).critical_regions}
eng = prog.engine()
eng.pruned_clear(); eng.frontier_root()
stat = {}
depth = max(eng.n_x, eng.n_t)
for lv in range(depth):
    gen = lv + 1 != depth
    st = eng.level_run(gen, keep_lowdim=True)
    cands, status = eng.frontier_get(), eng.level_status()
    for c, s in zip(cands.tolist(), status.tolist()):
        stat[tuple(c)] = s
    print('level', lv + 1, len(cands), numpy.bincount(status, minlength=6).tolist(), 'children', st.n_children)
    if not gen or st.n_children == 0: break
    eng.frontier_advance()
found = {k for k, s in stat.items() if s == 3}
missing = sorted(g - found, key=len)
print('graph', len(g), 'combinatorial', len(found), 'missing', len(missing))
for A in missing[:5]:
    print('missing', A, [ (A[:j], stat.get(A[:j], 'not generated')) for j in range(1, len(A) + 1)])

"""Per-level stage times of one generate_mpqp_data(n_x, n_theta, m, seed) program (second solve): python tools/dbg_levels.py nx nt m [seed=7] [max_levels]
(give max_levels for programs whose full depth is out of reach: the sweep's cells stop at 10^7 candidates)"""
import sys, warnings
sys.path.insert(0, '.')
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
nx, nt, mm = (int(v) for v in sys.argv[1:4])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 7
ml = int(sys.argv[5]) if len(sys.argv) > 5 else None
d = pg.generate_mpqp_data(nx, nt, mm, seed)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
m.solve(prog, max_levels=ml)
prof = []
sol = m.solve(prog, profile=prof, max_levels=ml)
print('regions', len(sol))
for p in prof:
    if p['depth'] > 0:
        print('k %2d n %7d wall %7.3f | kkt %6.3f theta %6.3f (items %d) x %6.3f xq %6.3f xqt %6.3f (%d of %d) x1 %6.3f region2 %6.3f (opt %d, side %d) | verdict %6.3f region %6.3f children %6.3f | status %s' % (
            p['k'], p['candidates'], p['ms_wall'], p['ms_kkt'], p['ms_theta'], p['n_theta_items'], p['ms_x'], p['ms_xq'], p['ms_xq_thread'], p['n_xq_thread'], p['n_xq_items'], p['ms_x1'], p['ms_region2'], p['n_opt'], p['region_side_stream'],
            p['ms_verdict'], p['ms_region'], p['ms_children'], p['status']))
        print('      x items %d, x1 %d, (x,theta) LPs %d, fallbacks to the LDS engine %d, pivots %d (quick test %d), dictionary bytes read %.2e written %.2e' % (
            p['n_x_items'], p['n_x1'], p['xtheta_lps'], p['xtheta_fallbacks'], p['lp_pivots'], p['xq_pivots'], p['dict_read_bytes'], p['dict_write_bytes']))

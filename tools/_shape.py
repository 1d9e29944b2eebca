import sys, time, warnings
sys.path.insert(0, '.')
warnings.simplefilter('ignore')
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
d = pg.generate_mpqp_data(14, 8, 30, 5)
prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
best = 1e9
for _ in range(4):
    prof = []
    t = time.perf_counter(); sol = m.solve(prog, max_levels=4, profile=prof); best = min(best, (time.perf_counter() - t) * 1e3)
print('solve %.2f ms, %d regions' % (best, len(sol.critical_regions)))
for p in prof:
    if p['depth'] > 0:
        print('  L%d n=%d wall %.3f | kkt %.3f theta %.3f x %.3f xq %.3f region2 %.3f n_opt %d' % (p['depth'], p['candidates'], p['ms_wall'], p['ms_kkt'], p['ms_theta'], p['ms_x'], p['ms_xq'], p['ms_region2'], p['n_opt']))

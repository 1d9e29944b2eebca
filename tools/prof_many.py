"""cProfile of solve_many on 128 small programs (where the host time of the shared launches goes): python tools/prof_many.py [n]"""
import cProfile, pstats, sys, time, warnings
sys.path.insert(0, '.')
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    progs = []
    for seed in range(n):
        d = pg.generate_mpqp_data(6, 3, 12, 5000 + seed)
        progs.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F']))
for p in progs:
    p.engine(0)
for _ in range(3):
    prof = []
    t = time.perf_counter(); sols = m.solve_many(progs, profile=prof); dt = time.perf_counter() - t
print('solve_many %.2f ms; device (shared launches) %.2f ms; levels %s' % (dt * 1e3, sum(p.get('ms_launches', 0) for p in prof), [(p['depth'], p['members'], round(p.get('ms_wall', 0), 2), round(p.get('ms_wait', 0), 2)) for p in prof]))
pr = cProfile.Profile(); pr.enable(); m.solve_many(progs); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)

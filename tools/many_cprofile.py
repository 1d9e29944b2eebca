"""cProfile of solve_many on 128 small programs (host side of the shared launches)."""
import cProfile, os, pstats, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
warnings.simplefilter('ignore')
progs = []
for seed in range(128):
    d = pg.generate_mpqp_data(6, 3, 12, 5000 + seed)
    progs.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F']))
for p in progs:
    p.engine(0)
for _ in range(2):
    mpqp_hip_combinatorial.solve_many(progs)
pr = cProfile.Profile(); pr.enable()
mpqp_hip_combinatorial.solve_many(progs)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)

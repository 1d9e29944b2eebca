import sys, warnings
sys.path.insert(0, '.')
import numpy
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as C
d = pg.generate_mpqp_data(7, 4, 24, 623692)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
for kw in ({'stream': False}, {'stream': True}):
    for rep in range(2):
        prof = []
        s = C.solve(prog, profile=prof, **kw)
        print(kw, len(s.critical_regions), [(p['candidates'], p['regions']) for p in prof])

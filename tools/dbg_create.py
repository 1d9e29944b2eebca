import os, sys, warnings
os.environ['MPC_DEBUG_CREATE'] = '1'
sys.path.insert(0, '.')
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
for (nx, nt, mm) in ((10, 2, 20), (10, 2, 30), (10, 6, 20)):
    d = pg.generate_mpqp_data(nx, nt, mm, 7)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
    print('program', nx, nt, mm, 'presolved n_c', prog.num_constraints(), 'A_t', prog.A_t.shape, flush=True)
    prof = []
    sol = m.solve(prog, profile=prof)
    print(' regions', len(sol), 'levels', [(p['k'], p['candidates'], round(p.get('ms_theta', 0), 3), round(p.get('ms_wall', 0), 3)) for p in prof if p['depth'] > 0], flush=True)

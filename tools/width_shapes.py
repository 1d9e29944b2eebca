"""The launch widths of round 4 against those of round 3 (MPC_TH_DIV=1 MPC_TH_MAXW=4 MPC_X2_WPC=16 MPC_X2_DIV=1 MPC_XQ_WPC=32) on programs of other
shapes than the bench configurations: best of five solves each (run once per setting; the switches are read when a handle is created)."""
import sys, time, warnings
sys.path.insert(0, '.')
warnings.simplefilter('ignore')
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
out = []
for (nx, nt, mm, seed, ml) in ((10, 6, 24, 3, 5), (14, 8, 30, 5, 4), (8, 10, 20, 7, 5), (16, 4, 26, 9, 5), (12, 3, 40, 11, 4)):
    d = pg.generate_mpqp_data(nx, nt, mm, seed)
    prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
    best, cands = 1e9, 0
    for _ in range(6):
        prof = []
        t = time.perf_counter(); sol = m.solve(prog, max_levels=ml, profile=prof); dt = (time.perf_counter() - t) * 1e3
        best = min(best, dt); cands = sum(p['candidates'] for p in prof)
    out.append('(%d,%d,%d): %d cand, %d regions, %.2f ms' % (nx, nt, mm, cands, len(sol.critical_regions), best))
    prog.release_engine()
print(' | '.join(out))

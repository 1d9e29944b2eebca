"""Many small programs at once: N random mpQPs (generate_mpqp_data(6, 3, 12, seed)) solved one after the other (solve) and together
(solve_many -> mpc_level_run_batch: one launch per stage and level for all of them).  usage: python tools/many_programs.py [N] [x t m]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppopt_amd import MPQP_Program, problem_generator as pg  # noqa: E402
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial  # noqa: E402

args = [int(v) for v in sys.argv[1:]]
N = args[0] if args else 128
x, t, m = (args[1:4] + [6, 3, 12][len(args[1:4]):])
warnings.simplefilter('ignore')
progs = []
for seed in range(N):
    d = pg.generate_mpqp_data(x, t, m, 5000 + seed)
    progs.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F']))
for p in progs:
    p.engine(0)
for rep in range(3):
    t0 = time.perf_counter()
    one = [mpqp_hip_combinatorial.solve(p) for p in progs]
    t1 = time.perf_counter()
    prof = []
    many = mpqp_hip_combinatorial.solve_many(progs, profile=prof)
    t2 = time.perf_counter()
    same = all(len(a.critical_regions) == len(b.critical_regions) for a, b in zip(one, many))
    print(f'rep {rep}: {N} programs ({x},{t},{m}), {sum(len(s) for s in one)} regions, {sum(p["candidates"] for p in prof)} candidates: one by one {1e3*(t1-t0):.1f} ms '
          f'({1e3*(t1-t0)/N:.2f} per program), together {1e3*(t2-t1):.1f} ms ({1e3*(t2-t1)/N:.3f} per program), x{(t1-t0)/(t2-t1):.1f}; same region counts {same}; '
          f'device {sum(p.get("ms_launches", 0) for p in prof):.1f} ms, levels {len(prof) - 1}, shared {sum(p["shared_launches"] for p in prof)}/{sum(p["members"] for p in prof)}')

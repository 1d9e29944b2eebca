"""Timing of the mixed-integer enumeration (solve_mpmiqp) on a synthetic mixed-integer mpQP, with the host/device split
per stage.  usage: python tools/mi_run.py [x t m n_bin seed]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ppopt_amd import MPMIQP_Program  # noqa: E402
from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp  # noqa: E402
from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp  # noqa: E402
from ppopt_amd.problem_generator import generate_mpmiqp_data  # noqa: E402

args = [int(v) for v in sys.argv[1:]]
x, t, m, nb, seed = (args + [6, 3, 12, 5, 0][len(args):])[:5]
d = generate_mpmiqp_data(x, t, m, nb, seed)
warnings.simplefilter('ignore')
for rep in range(3):
    t0 = time.perf_counter()
    prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
    t1 = time.perf_counter()
    combos = prog.feasible_combinations()
    t2 = time.perf_counter()
    subs = [prog.generate_substituted_problem(f) for f in combos]
    t3 = time.perf_counter()
    sols = [solve_mpqp(s, mpqp_algorithm.combinatorial) for s in subs]
    t4 = time.perf_counter()
    n_reg = sum(len(s) for s in sols)
    print(f'rep {rep}: presolve {1e3*(t1-t0):.1f} ms, fixations {1e3*(t2-t1):.1f} ms ({len(combos)} of {1 << nb} feasible), '
          f'substitute+presolve {1e3*(t3-t2):.1f} ms, solve {1e3*(t4-t3):.1f} ms, {n_reg} regions; '
          f'n_c of subs {sorted(set(s.num_constraints() for s in subs))}')
for cores in (1, 1, 2, 4, 8, 8):
    t0 = time.perf_counter()
    sol = solve_mpmiqp(prog, num_cores=cores)
    dt = time.perf_counter() - t0
    print(f'solve_mpmiqp(num_cores={cores}): {1e3*dt:.1f} ms, {len(sol)} regions, {len(combos)/dt:.0f} sub-programs/s')

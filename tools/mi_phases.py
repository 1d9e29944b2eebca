"""Phases of the default mixed-integer enumeration (bench workload): substitution + presolve + set-up of the sub-programs, the shared
solve, collecting the regions.  python tools/mi_phases.py"""
import sys, time, warnings
sys.path.insert(0, '.')
from ppopt_amd import MPMIQP_Program
from ppopt_amd.mp_solvers import mpmiqp_enumeration, mpqp_hip_combinatorial
from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
from ppopt_amd.problem_generator import generate_mpmiqp_data
d = generate_mpmiqp_data(8, 4, 16, 6, 1)
warnings.simplefilter('ignore')
prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
real = mpqp_hip_combinatorial.solve_many
marks = {}
def timed(*a, **k):
    marks['t_in'] = time.perf_counter()
    out = real(*a, **k)
    marks['t_out'] = time.perf_counter()
    return out
mpqp_hip_combinatorial.solve_many = timed
for _ in range(2):
    solve_mpmiqp(prog)
for rep in range(4):
    t0 = time.perf_counter(); sol = solve_mpmiqp(prog); t1 = time.perf_counter()
    print('total %.1f ms: before solve_many %.1f, solve_many %.1f, after %.1f; %d regions' % (1e3 * (t1 - t0), 1e3 * (marks['t_in'] - t0), 1e3 * (marks['t_out'] - marks['t_in']), 1e3 * (t1 - marks['t_out']), len(sol)))

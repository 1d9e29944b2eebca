"""cProfile of the default mixed-integer enumeration (bench workload): where the host time outside the shared solve goes.  python tools/mi_cprofile.py"""
import cProfile, pstats, sys, warnings
sys.path.insert(0, '.')
from ppopt_amd import MPMIQP_Program
from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
from ppopt_amd.problem_generator import generate_mpmiqp_data
d = generate_mpmiqp_data(8, 4, 16, 6, 1)
warnings.simplefilter('ignore')
prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
for _ in range(2):
    solve_mpmiqp(prog)
pr = cProfile.Profile(); pr.enable(); solve_mpmiqp(prog); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)

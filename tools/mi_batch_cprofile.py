"""cProfile of the batched mixed-integer enumeration (bench workload), host side."""
import cProfile
import os
import pstats
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppopt_amd import MPMIQP_Program  # noqa: E402
from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp  # noqa: E402
from ppopt_amd.problem_generator import generate_mpmiqp_data  # noqa: E402

d = generate_mpmiqp_data(8, 4, 16, 6, 1)
warnings.simplefilter('ignore')
prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
for _ in range(2):
    solve_mpmiqp(prog, num_cores=1)
pr = cProfile.Profile()
pr.enable()
sol = solve_mpmiqp(prog, num_cores=1)
pr.disable()
print(len(sol), 'regions')
pstats.Stats(pr).sort_stats('tottime').print_stats(28)

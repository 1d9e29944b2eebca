"""cProfile of the mixed-integer enumeration on the bench's extra workload (run on the GPU box)."""
import sys, time, cProfile, pstats, warnings
sys.path.insert(0, '.')
from ppopt_amd import MPMIQP_Program
from ppopt_amd.problem_generator import generate_mpmiqp_data
from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
d = generate_mpmiqp_data(8, 4, 16, n_bin=6, seed=1)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
cores = int(sys.argv[1]) if len(sys.argv) > 1 else 8
solve_mpmiqp(prog, num_cores=cores)
pr = cProfile.Profile(); pr.enable(); t = time.perf_counter(); s = solve_mpmiqp(prog, num_cores=cores); dt = time.perf_counter() - t; pr.disable()
print(len(s.critical_regions), 'regions', dt * 1e3, 'ms')
pstats.Stats(pr).sort_stats('tottime').print_stats(18)

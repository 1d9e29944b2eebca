"""The mixed-integer enumeration of bench.py (generate_mpmiqp_data(8,4,16,6,1), 64 fixations, shared launches) a few times, for a
rocprofv3 --kernel-trace --stats run:  rocprofv3 --kernel-trace --stats -d /tmp/mi -o run -- python3 tools/mi_profile.py [reps]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppopt_amd import MPMIQP_Program
from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
from ppopt_amd.problem_generator import generate_mpmiqp_data
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = generate_mpmiqp_data(8, 4, 16, 6, 1)
warnings.simplefilter('ignore')
p = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
solve_mpmiqp(p)
ts = []
for _ in range(reps):
    t = time.perf_counter(); s = solve_mpmiqp(p); ts.append(1e3 * (time.perf_counter() - t))
print('mi enumeration: best %.1f ms, all %s; regions %d' % (min(ts), [round(v, 1) for v in ts], len(s)))

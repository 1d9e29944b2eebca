import sys
sys.path.insert(0, '.')
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
mpqp_hip_combinatorial.solve(prog, max_levels=ml)

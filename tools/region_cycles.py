import sys, os
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
prog = bench.build_program('c4', 0)
mpqp_hip_combinatorial.solve(prog, max_levels=5)
print('---- second solve', file=sys.stderr, flush=True)
mpqp_hip_combinatorial.solve(prog, max_levels=5)

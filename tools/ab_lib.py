"""A/B of two builds of the library on the bench workload: python tools/ab_lib.py <lib.so> [workload]"""
import sys
sys.path.insert(0, '.')
from ppopt_amd import _lib
_lib.LIB_PATH = sys.argv[1]
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
wl = sys.argv[2] if len(sys.argv) > 2 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
import time
warm = [mpqp_hip_combinatorial.solve(prog, max_levels=ml) for _ in range(3)]
del warm
best = None
for _ in range(5):
    t = time.perf_counter(); prof = []; sol = mpqp_hip_combinatorial.solve(prog, max_levels=ml, profile=prof); dt = time.perf_counter() - t
    row = (dt * 1e3, sum(p.get('ms_verdict', 0) for p in prof), sum(p.get('ms_region', 0) for p in prof), sum(p.get('ms_theta', 0) for p in prof), sum(p.get('ms_x', 0) for p in prof), sum(p.get('ms_region2', 0) for p in prof))
    best = row if best is None or row[0] < best[0] else best
print(sys.argv[1], 'wall %.2f verdict %.2f region %.2f | theta %.2f x2 %.2f region2 %.2f' % best, 'regions', len(sol.critical_regions))

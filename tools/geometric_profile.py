import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_geometric as GE
prog = bench.build_program('c4')
GE.solve(prog, max_regions=5000)
pr = cProfile.Profile(); pr.enable(); t = time.perf_counter(); s = GE.solve(prog); dt = time.perf_counter() - t; pr.disable()
print(len(s.critical_regions), dt)
pstats.Stats(pr).sort_stats('tottime').print_stats(16)

import sys, time, cProfile, pstats, warnings
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
prog = bench.build_program('c4')
for _ in range(2): mpqp_hip_combinatorial.solve(prog, max_levels=5)
t=time.perf_counter(); prof=[]; sol = mpqp_hip_combinatorial.solve(prog, max_levels=5, profile=prof); dt=time.perf_counter()-t
print('solve wall %.1f ms; kernel ms %.1f' % (dt*1e3, sum(p.get('ms_verdict',0)+p.get('ms_region',0)+p.get('ms_children',0) for p in prof)))
for p in prof: print(p.get('depth'), p.get('candidates'), 'wall %.1f' % p.get('ms_wall', 0), 'v/r/c %.1f/%.1f/%.1f' % (p.get('ms_verdict',0), p.get('ms_region',0), p.get('ms_children',0)))
pr = cProfile.Profile(); pr.enable(); mpqp_hip_combinatorial.solve(prog, max_levels=5); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)

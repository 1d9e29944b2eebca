"""Host-side profile of one steady-state solve of the bench workload (run on the GPU box)."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
warm = [mpqp_hip_combinatorial.solve(prog, max_levels=ml) for _ in range(3)]
del warm
sol = None
for _ in range(3):
    t = time.perf_counter(); prof = []; sol = mpqp_hip_combinatorial.solve(prog, max_levels=ml, profile=prof); dt = time.perf_counter() - t
    print('solve wall %.1f ms; kernel ms %.1f' % (dt * 1e3, sum(p.get('ms_verdict', 0) + p.get('ms_region', 0) + p.get('ms_children', 0) for p in prof)))
for p in prof:
    print(p.get('depth'), p.get('candidates'), 'wall %.1f' % p.get('ms_wall', 0), 'v/r/c %.1f/%.1f/%.1f' % (p.get('ms_verdict', 0), p.get('ms_region', 0), p.get('ms_children', 0)))
pr = cProfile.Profile(); pr.enable(); sol = mpqp_hip_combinatorial.solve(prog, max_levels=ml); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)

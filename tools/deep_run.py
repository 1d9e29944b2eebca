"""Config 4 one level deeper than the bench (6.5e6 candidates, 9.4 GB of cached dictionaries): a scale check."""
import sys, time
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 6
prog = bench.build_program('c4')
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    prof = []
    t = time.perf_counter()
    sol = mpqp_hip_combinatorial.solve(prog, max_levels=levels, profile=prof)
    dt = time.perf_counter() - t
    n = sum(p['candidates'] for p in prof)
    print(f'run {rep}: {n} candidates, {len(sol.critical_regions)} regions, {dt * 1e3:.1f} ms, {n / dt / 1e6:.1f} M candidates/s')
for p in prof:
    print(p['depth'], p['candidates'], p['status'], 'regions', p['regions'], 'ms v/r/c %.1f/%.1f/%.1f wall %.2f | theta %.2f x %.2f region2 %.2f' % (p.get('ms_verdict', 0), p.get('ms_region', 0), p.get('ms_children', 0), p.get('ms_wall', 0), p.get('ms_theta', 0), p.get('ms_x', 0), p.get('ms_region2', 0)))

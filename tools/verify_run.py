"""End-to-end validation of a solved workload without any oracle: every region's laws satisfy the KKT conditions of the
program at the region's Chebyshev centre (all centres from one LP batch on the device) and point location returns that
region there.  usage: python tools/verify_run.py [c4|c3|c2]"""
import sys, time
sys.path.insert(0, '.')
import numpy
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
prog = bench.build_program(wl)
sol = mpqp_hip_combinatorial.solve(prog, max_levels=bench.WORKLOADS[wl][2])
t0 = time.perf_counter(); centres, radii = sol.chebyshev_centres(); t1 = time.perf_counter()
worst = {}
for r, c in zip(sol.critical_regions, centres):
    for k, v in sol.kkt_residuals(r, c).items():
        worst[k] = max(worst.get(k, 0.0), v)
t2 = time.perf_counter()
ok = sol.verify_solution()
print(f'{wl}: {len(sol)} regions, Chebyshev LPs {1e3*(t1-t0):.1f} ms, smallest radius {numpy.nanmin(radii):.3e}, worst KKT residuals {worst}, '
      f'residual loop {t2-t1:.2f} s, verify_solution {ok}')

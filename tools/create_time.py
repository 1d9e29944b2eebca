import sys, time
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
for wl in ('c4', 'c2'):
    prog = bench.build_program(wl)
    ml = bench.WORKLOADS[wl][2]
    t0 = time.perf_counter(); eng = prog.engine(0); t1 = time.perf_counter()
    print(wl, 'engine create %.2f ms' % ((t1 - t0) * 1e3))
    prog.release_engine()
    t0 = time.perf_counter(); eng = prog.engine(0); t1 = time.perf_counter()
    print(wl, 'engine create again %.2f ms' % ((t1 - t0) * 1e3))
    for i in range(4):
        t0 = time.perf_counter(); s = mpqp_hip_combinatorial.solve(prog, max_levels=ml); t1 = time.perf_counter()
        print(wl, 'solve %d: %.2f ms' % (i, (t1 - t0) * 1e3), len(s.critical_regions))

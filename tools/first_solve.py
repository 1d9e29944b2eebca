import sys, time
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
for wl in ('c4', 'c2'):
    ml = bench.WORKLOADS[wl][2]
    warm = bench.build_program(wl); m.solve(warm, max_levels=ml); m.solve(warm, max_levels=ml)     # modules loaded, pools warm
    for rep in range(3):
        prog = bench.build_program(wl)
        t0 = time.perf_counter(); eng = prog.engine(0, closed=True); t1 = time.perf_counter()
        times = []
        for i in range(4):
            t = time.perf_counter(); m.solve(prog, max_levels=ml); times.append((time.perf_counter() - t) * 1e3)
        print(wl, 'create %.2f ms; solves of a fresh handle: %s ms' % ((t1 - t0) * 1e3, [round(x, 2) for x in times]))
        prog.release_engine()

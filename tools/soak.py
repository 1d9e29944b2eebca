"""Soak: many solves of one program in a row; host RSS and the time per solve must stay flat.  python tools/soak.py [workload] [n]"""
import os, sys, time
sys.path.insert(0, '.')
import psutil
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
prog = bench.build_program(wl)
ml = bench.WORKLOADS[wl][2]
proc = psutil.Process(os.getpid())
for _ in range(20):
    m.solve(prog, max_levels=ml)
rss0 = proc.memory_info().rss
marks = []
t0 = time.perf_counter()
for i in range(n):
    sol = m.solve(prog, max_levels=ml)
    if (i + 1) % (n // 5) == 0:
        marks.append((i + 1, (time.perf_counter() - t0) / (i + 1) * 1e3, (proc.memory_info().rss - rss0) / 1e6))
print(wl, 'regions', len(sol.critical_regions), '; after n solves: (n, mean ms per solve so far, RSS growth MB):', [(a, round(b, 3), round(c, 1)) for a, b, c in marks])

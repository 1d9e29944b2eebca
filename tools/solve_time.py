"""Wall time of solve() as a user runs it (no profile: no HIP-event records inside the levels): python tools/solve_time.py [workload=c4] [reps=30]
prints best / median ms per solve."""
import sys, time, statistics
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
for _ in range(5):
    m.solve(prog, max_levels=ml)
ts = []
sol = None
for _ in range(reps):
    sol = None
    t = time.perf_counter(); sol = m.solve(prog, max_levels=ml); ts.append((time.perf_counter() - t) * 1e3)
print('%s solve: best %.3f ms, median %.3f ms over %d; regions %d' % (wl, min(ts), statistics.median(ts), reps, len(sol.critical_regions)))

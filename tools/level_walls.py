"""Per-level wall times of the in-library solve loop (best of N solves): python tools/level_walls.py [workload] [reps] [max_levels]"""
import sys, time
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
ml = int(sys.argv[3]) if len(sys.argv) > 3 else bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
for _ in range(4):
    m.solve(prog, max_levels=ml)
best = None
for _ in range(reps):
    prof = []
    t = time.perf_counter(); sol = m.solve(prog, max_levels=ml, profile=prof); dt = (time.perf_counter() - t) * 1e3
    lv = [p for p in prof if p['depth'] > 0]
    row = (dt, [round(p.get('ms_wall', 0), 3) for p in lv], [round(p.get('ms_xq', 0) + p.get('ms_x', 0), 3) for p in lv],
           [round(p.get('ms_region2', 0), 3) for p in lv], [round(p.get('ms_theta', 0), 3) for p in lv], [round(p.get('ms_kkt', 0), 3) for p in lv],
           [(round(p.get('ms_xq_thread', 0), 3), p.get('n_xq_thread', 0), p.get('n_xq_items', 0)) for p in lv if p.get('n_xq_items', 0)])
    if best is None or row[0] < best[0]:
        best = row
print('%s solve %.3f ms; level walls %s (sum %.3f); x stage %s; region2 %s; theta %s; kkt %s; xq thread pass (ms, decided, of) %s; regions %d' % (wl, best[0], best[1], sum(best[1]), best[2], best[3], best[4], best[5], best[6], len(sol.critical_regions)))

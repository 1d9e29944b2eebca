"""Wall time of mpqp_hip_combinatorial.solve per level (profile['ms_wall']) against the whole call (run on the GPU box):
python tools/solve_split.py [c4|c3|c2]"""
import sys, time, gc
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
prog = bench.build_program(wl)
ml = bench.WORKLOADS[wl][2]
for _ in range(5):
    mpqp_hip_combinatorial.solve(prog, max_levels=ml)
gc.collect(); gc.freeze()
for _ in range(3):
    prof = []
    t0 = time.perf_counter()
    s = mpqp_hip_combinatorial.solve(prog, max_levels=ml, profile=prof)
    tot = (time.perf_counter() - t0) * 1e3
    lv = [round(p['ms_wall'], 3) for p in prof if p.get('depth', 0) > 0]
    print('total %.3f ms | levels %s sum %.3f | rest %.3f' % (tot, lv, sum(lv), tot - sum(lv)))

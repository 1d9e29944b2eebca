"""What a bench step costs beyond the in-library solve loop: python tools/step_overhead.py [workload]"""
import gc, sys, time
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
for _ in range(5):
    sol = m.solve(prog, max_levels=ml)
pc = time.perf_counter
def run(n, keep, collect):
    kept = []
    t0 = pc()
    for _ in range(n):
        s = m.solve(prog, max_levels=ml)
        if keep:
            kept.append(s)
    dt = (pc() - t0) / n * 1e3
    return dt
for label, keep, gcoff in (('drop previous', False, False), ('keep all', True, False), ('drop, gc disabled', False, True), ('keep, gc disabled', True, True)):
    if gcoff:
        gc.disable()
    print('%-22s %.3f ms per solve' % (label, run(20, keep, gcoff)))
    gc.enable(); gc.collect()
t0 = pc(); del sol; print('freeing one solution: %.3f ms' % ((pc() - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    s = m.solve(prog, max_levels=ml)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)

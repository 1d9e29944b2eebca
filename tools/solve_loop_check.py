"""In-library level loop (mpc_solve_start) against the level-by-level loop of round 3: same regions, times of both.
python tools/solve_loop_check.py [workload ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m

for wl in (sys.argv[1:] or ['c4', 'c3', 'c2']):
    ml = bench.WORKLOADS[wl][2]
    prog = bench.build_program(wl)
    out = {}
    for mode in (True, False, True, False):
        m.SOLVE_LOOP = mode
        warm = [m.solve(prog, max_levels=ml) for _ in range(3)]
        del warm
        best = 1e9
        for _ in range(7):
            t = time.perf_counter(); prof = []; sol = m.solve(prog, max_levels=ml, profile=prof); best = min(best, time.perf_counter() - t)
        out.setdefault(mode, []).append(best * 1e3)
        key = sorted((tuple(cr.active_set), cr.E.shape, float(numpy.sum(cr.A)), float(numpy.sum(cr.E)), tuple(cr.omega_set), tuple(cr.lambda_set)) for cr in sol.critical_regions)
        out.setdefault('key%d' % mode, key)
        out.setdefault('prof%d' % mode, [(p['candidates'], p['status'], p['regions']) for p in prof])
    same = out['key1'] == out['key0'] and out['prof1'] == out['prof0']
    print(wl, 'in-library loop ms', ['%.2f' % v for v in out[True]], 'level-by-level ms', ['%.2f' % v for v in out[False]], 'regions', len(out['key1']), 'identical', same, flush=True)
    prog.release_engine()

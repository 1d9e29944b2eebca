"""Host-side split of the in-library solve loop: per level, when its records arrive, how long this thread waits for chunks and how long it
builds objects; then the epilogue.  python tools/solve_trace.py [workload]"""
import sys, time
sys.path.insert(0, '.')
import numpy
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
from ppopt_amd.region_batch import RegionBatch
from ppopt_amd.solution import Solution

wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
for _ in range(4):
    m.solve(prog, max_levels=ml)
eng = prog.engine(0, closed=True)
pc = time.perf_counter
rows = []
for rep in range(5):
    T0 = pc()
    twin = eng.twin(); twin.level_start(False, only_base=True)
    t_twin = pc()
    eng.solve_start(ml, stream=True, fetch=True)
    t_start = pc()
    regions = []
    lv = 0
    per = []
    while True:
        a = pc(); info = eng.solve_level(lv); b = pc()
        if info is None:
            break
        mode, k, n, hd, hi, er, chunk, n_chunks = info
        c = pc()
        w = bld = 0.0
        if mode == 1:
            batch = RegionBatch(hd, hi, er, eng.n_x, eng.n_t, eng.n_c, eng.n_tc, k, ())
            sc = hi[:, 0]
            for j in range(n_chunks):
                x = pc(); eng.solve_chunk_wait(lv, j); y = pc()
                lo = j * chunk
                regions.extend(batch.regions_of((lo + numpy.flatnonzero(sc[lo:lo + chunk] == 3)).tolist()))
                z = pc(); w += y - x; bld += z - y
        elif mode == 2:
            y = pc(); regions.extend(RegionBatch(hd, hi, er, eng.n_x, eng.n_t, eng.n_c, eng.n_tc, k, numpy.flatnonzero(hi[:, 0] == 3)).regions()); bld += pc() - y
        d = pc(); st, msw = eng.solve_level_wait(lv); e = pc()
        per.append((lv + 1, mode, n, int(st.n_regions), (a - T0) * 1e3, (b - a) * 1e3, (c - b) * 1e3, w * 1e3, bld * 1e3, (e - d) * 1e3, msw, (e - T0) * 1e3))
        lv += 1
    t_loop = pc()
    eng.solve_wait(); t_sw = pc()
    twin.level_wait(); res = twin.base_result(); t_base = pc()
    rows.append((per, (t_twin - T0) * 1e3, (t_start - t_twin) * 1e3, (t_loop - T0) * 1e3, (t_sw - t_loop) * 1e3, (t_base - t_sw) * 1e3, (t_base - T0) * 1e3))
per, a, b, c, d, e, f = rows[-1]
print('twin start %.3f  solve_start %.3f  loop end at %.3f  solve_wait %.3f  base %.3f  total %.3f ms' % (a, b, c, d, e, f))
print('level mode      n  regions | asked at  wait info  adopt  chunk-wait  build  level_wait | level wall  done at')
for r in per:
    print('%5d %4d %7d %7d | %8.3f %9.3f %6.3f %10.3f %6.3f %10.3f | %9.3f %8.3f' % r)
print('totals of 5 reps:', ['%.2f' % r[-1] for r in rows])

"""Wall-time split of one steady-state STREAMED solve (mpc_level_start / stream_info / chunk_wait / level_wait), per level
(run on the GPU box)."""
import sys, time
sys.path.insert(0, '.')
import numpy
import bench, gc
from ppopt_amd.region_batch import RegionBatch
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
eng = prog.engine(0)


def run():
    rows = []
    regs = []
    eng.pruned_clear(); eng.frontier_root()
    depth = 0
    t00 = time.perf_counter()
    while True:
        depth += 1
        gen = (ml is None) or depth != ml
        t0 = time.perf_counter()
        eng.level_start(gen, True)
        t1 = time.perf_counter()
        info = eng.level_stream_info()
        t2 = time.perf_counter()
        t_wait = t_obj = 0.0
        first = last = 0.0
        if info is not None:
            hd, hi, er, chunk, n_chunks = info
            batch = RegionBatch(hd, hi, er, eng.n_x, eng.n_t, eng.n_c, eng.n_tc, eng.frontier_info()[1], ())
            col = hi[:, 0]
            for j in range(n_chunks):
                a = time.perf_counter(); eng.level_chunk_wait(j); b = time.perf_counter()
                if j == 0:
                    first = b - t2
                lo = j * chunk
                regs.extend(batch.regions_of((lo + numpy.flatnonzero(col[lo:lo + chunk] == 3)).tolist()))
                c = time.perf_counter()
                t_wait += b - a; t_obj += c - b
            last = time.perf_counter() - t2
        t3 = time.perf_counter()
        st = eng.level_wait()
        t4 = time.perf_counter()
        rows.append((depth, int(st.n), int(st.n_opt), round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3), round(first * 1e3, 3), round(t_wait * 1e3, 3),
                     round(t_obj * 1e3, 3), round((t4 - t3) * 1e3, 3), round((t4 - t0) * 1e3, 3),
                     round(st.ms_verdict + st.ms_region + st.ms_children, 3)))
        if not gen or st.n_children == 0:
            break
        eng.frontier_advance()
    return rows, (time.perf_counter() - t00) * 1e3, len(regs)


keep = [run() for _ in range(3)]
del keep
gc.collect(); gc.freeze()
for _ in range(3):
    rows, tot, n = run()
    print('total %.2f ms, %d regions' % (tot, n))
    print('depth n n_opt | start info(first-stage wait) first-chunk chunk-waits objects level_wait | level wall | kernel ms')
    for r in rows:
        print(r)

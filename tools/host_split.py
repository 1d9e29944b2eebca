"""Wall-time split of one steady-state solve: level_run / region fetch / region objects (run on the GPU box)."""
import sys, time
sys.path.insert(0, '.')
import bench, gc
from ppopt_amd.region_batch import RegionBatch
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
eng = prog.engine(0)
def run():
    t = {'run': 0.0, 'fetch': 0.0, 'objects': 0.0, 'advance': 0.0}
    regs = []
    eng.pruned_clear(); eng.frontier_root()
    depth = 0
    t00 = time.perf_counter()
    while True:
        depth += 1
        gen = (ml is None) or depth != ml
        t0 = time.perf_counter(); st = eng.level_run(gen); t1 = time.perf_counter(); t['run'] += t1 - t0
        if st.n_regions:
            hd, hi, er, kk, slots = eng.level_regions_slots(); t2 = time.perf_counter(); t['fetch'] += t2 - t1
            regs.extend(RegionBatch(hd, hi, er, eng.n_x, eng.n_t, eng.n_c, eng.n_tc, kk, slots).regions()); t3 = time.perf_counter(); t['objects'] += t3 - t2
        if not gen or st.n_children == 0:
            break
        t4 = time.perf_counter(); eng.frontier_advance(); t['advance'] += time.perf_counter() - t4
    t['total'] = time.perf_counter() - t00
    return t, regs
keep = [run() for _ in range(3)]
del keep
gc.collect(); gc.freeze()
for _ in range(3):
    t, regs = run()
    print({k: round(v * 1e3, 2) for k, v in t.items()}, len(regs))

"""Wall-time split of a complete graph solve with device bookkeeping (run on the GPU box): python tools/graph_split.py [c4|c3]"""
import sys, time
sys.path.insert(0, '.')
import numpy, gc
import bench
from ppopt_amd.region_batch import RegionBatch
from ppopt_amd.mp_solvers import mpqp_hip_combi_graph as G
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
prog = bench.build_program(wl)
eng = prog.engine()
seeds = G._sets_to_masks(G._seed_active_sets(prog, eng), eng.mask_words)


def run():
    t = {'wave': 0.0, 'group_run': 0.0, 'fetch': 0.0, 'objects': 0.0, 'close': 0.0, 'kernel_ms': 0.0}
    regs = []
    t0 = time.perf_counter()
    eng.graph_begin(seeds, 1)
    while True:
        a = time.perf_counter(); groups, _ = eng.graph_wave(); t['wave'] += time.perf_counter() - a
        if not groups:
            break
        for gi in range(len(groups)):
            a = time.perf_counter(); st = eng.graph_group_run(gi); b = time.perf_counter(); t['group_run'] += b - a
            t['kernel_ms'] += st.ms_verdict + st.ms_region + st.ms_children
            if st.n_regions:
                hd, hi, er, kk, slots = eng.level_regions_slots(); c = time.perf_counter(); t['fetch'] += c - b
                regs.extend(RegionBatch(hd, hi, er, eng.n_x, eng.n_t, eng.n_c, eng.n_tc, kk, slots).regions()); t['objects'] += time.perf_counter() - c
        a = time.perf_counter(); eng.graph_wave_close(); t['close'] += time.perf_counter() - a
    t['total'] = time.perf_counter() - t0
    return t, len(regs)


keep = run(); del keep
gc.collect(); gc.freeze()
for _ in range(3):
    t, n = run()
    print({k: round(v * (1 if k == 'kernel_ms' else 1e3), 1) for k, v in t.items()}, n)

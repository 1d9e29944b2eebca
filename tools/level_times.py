"""Per-level device and wall times of one steady-state solve in the synchronous loop (run on the GPU box):
python tools/level_times.py [c4|c3|c2]"""
import sys, time
sys.path.insert(0, '.')
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
eng = prog.engine(0)


def run(show):
    eng.pruned_clear(); eng.frontier_root()
    depth = 0
    while True:
        depth += 1
        gen = (ml is None) or depth != ml
        t0 = time.perf_counter(); st = eng.level_run(gen); w = time.perf_counter() - t0
        if show:
            print(f'k={st.k} n={st.n} opt={st.n_opt} regions={st.n_regions} wall {w*1e3:.3f} ms | events: verdict {st.ms_verdict:.3f} region {st.ms_region:.3f} '
                  f'children {st.ms_children:.3f} | kernels: theta {st.ms_theta:.3f} x {st.ms_x:.3f} region2 {st.ms_region2:.3f}')
        if not gen or st.n_children == 0:
            break
        eng.frontier_advance()


for i in range(6):
    run(i >= 4)
    if i >= 4:
        print()

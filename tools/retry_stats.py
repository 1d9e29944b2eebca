"""Per-level retry statistics of the bench workload (run on the GPU box)."""
import sys
sys.path.insert(0, '.')
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
prog = bench.build_program(wl)
eng = prog.engine(0)
eng.pruned_clear(); eng.frontier_root()
depth = 0
while True:
    depth += 1
    gen = (ml is None) or depth != ml
    st = eng.level_run(gen)
    print(f'k={st.k} n={st.n} status={list(st.n_status)} xlp={st.n_xtheta_lp} verdict_retry={st.n_xtheta_fallback} region_retry={st.n_region_retry} '
          f'cached={st.n_x_cached} pivots={st.lp_pivots} ms v/r/c {st.ms_verdict:.2f}/{st.ms_region:.2f}/{st.ms_children:.2f}')
    if not gen or st.n_children == 0:
        break
    eng.frontier_advance()

"""Explores how many candidates the full combinatorial tree of generated problems has (to pick bench workloads)."""
import sys, time, warnings
sys.path.insert(0, '.')
import numpy
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial

CAP = float(sys.argv[1]) if len(sys.argv) > 1 else 3e6
configs = [('rand', 20, 8, 20, s) for s in range(4)] + [('rand', 20, 8, 12, s) for s in range(3)] + \
          [('rand', 16, 6, 16, s) for s in range(3)] + [('qt', 10), ('qt', 6), ('qt', 8), ('dbl', 8), ('dbl', 10)]
for cfg in configs:
    if cfg[0] == 'rand':
        d = pg.generate_mpqp_data(*cfg[1:])
    elif cfg[0] == 'qt':
        d = pg.quad_tank_data(cfg[1])
    else:
        d = pg.double_integrator_data(cfg[1])
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=d['equality_indices'])
    tp = time.time() - t0
    eng = prog.engine()
    max_depth = max(eng.n_x, eng.n_t) - eng.n_eq
    eng.pruned_clear(); eng.frontier_root()
    tot = 0; reg = 0; ms = 0.0; counts = []
    t0 = time.time(); done = True
    for depth in range(max_depth):
        gen = depth + 1 != max_depth
        st = eng.level_run(gen)
        tot += st.n; reg += st.n_regions; ms += st.ms_total; counts.append(int(st.n))
        if not gen or st.n_children == 0: break
        if st.n_children > CAP: done = False; break
        eng.frontier_advance()
    print(cfg, f'n_x {eng.n_x} n_t {eng.n_t} n_c {eng.n_c} e {eng.n_eq} n_tc {eng.n_tc} presolve {tp:.2f}s | complete={done} levels={counts} total {tot} regions {reg} kernel {ms:.1f} ms wall {time.time()-t0:.2f}s', flush=True)
    eng.close()

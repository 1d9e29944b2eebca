"""The drivers on MPC programs (equality rows, degenerate vertices): double integrator N = 2..7, quad tank N = 2..6 (run on the GPU box)."""
import sys, warnings, time
sys.path.insert(0, '.')
import numpy
from ppopt_amd import MPQP_Program, problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as C, mpqp_hip_combi_graph as G, mpqp_hip_geometric as GE
cases = [('dblint', n) for n in range(2, 8)] + [('quadtank', n) for n in range(2, 7)]
for kind, n in cases:
    d = pg.double_integrator_data(n) if kind == 'dblint' else pg.quad_tank_data(n)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=d['equality_indices'])
    depth = max(prog.num_x(), prog.num_t()) - len(prog.equality_indices)
    line = f'{kind} N={n}: n_x {prog.num_x()} n_c {prog.num_constraints()} n_eq {len(prog.equality_indices)}'
    ref = None
    if prog.num_constraints() <= 40 or depth <= 6:
        t = time.perf_counter(); ref = {tuple(r.active_set) for r in C.solve(prog).critical_regions}; line += f' | combinatorial {len(ref)} ({(time.perf_counter() - t) * 1e3:.0f} ms)'
    out = {}
    for label, fn in (('graph', G.solve_graph), ('geometric', GE.solve)):
        t = time.perf_counter()
        try:
            s = fn(prog)
            out[label] = {tuple(r.active_set) for r in s.critical_regions}
            line += f' | {label} {len(out[label])} ({(time.perf_counter() - t) * 1e3:.0f} ms)'
        except Exception as e:
            line += f' | {label}: {type(e).__name__} {str(e)[:80]}'
    if 'graph' in out and 'geometric' in out:
        a, b = out['graph'], out['geometric']
        line += f' | graph-only {len(a - b)} geometric-only {len(b - a)}'
        if ref is not None:
            line += f' | vs combinatorial: graph -{len(ref - a)} +{len(a - ref)}, geometric -{len(ref - b)} +{len(b - ref)}'
    print(line, flush=True)
    prog.release_engine()

import sys, warnings, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy
from conftest import load_golden
from test_host_logic import build_program
from ppopt_amd import Solver
from ppopt_amd.mp_solvers import mpqp_hip_combi_graph as G, mpqp_hip_combinatorial as C
for name in ['c5_control_allocation', 'c3_quadtank_n10']:
    g = load_golden(name)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = build_program(g, Solver())
    if name.startswith('c5'):
        t = time.perf_counter(); ref = C.solve(prog); dt = time.perf_counter() - t
        rk = {tuple(r.active_set) for r in ref.critical_regions}
        print(name, 'combinatorial (complete):', len(rk), 'regions in %.1f ms' % (dt * 1e3))
    else:
        rk = None
    for label, fn in (('graph', G.solve_graph), ('combinatorial_graph', G.solve)):
        prof = []
        t = time.perf_counter(); s = fn(prog, profile=prof, max_candidates=50000000); dt = time.perf_counter() - t
        k = {tuple(r.active_set) for r in s.critical_regions}
        print(name, label, len(k), 'regions,', sum(p['candidates'] for p in prof), 'sets, %.1f ms' % (dt * 1e3), 'complete' if s.is_complete else 'capped',
              '' if rk is None else 'missing %d extra %d' % (len(rk - k), len(k - rk)))

"""The connected-graph traversal (mpqp_algorithm.combinatorial_graph) on a bench workload, complete solution (run on the GPU
box): python tools/graph_run.py [c3|c2|c4] [max_candidates] [combinatorial_graph|graph]"""
import sys, time
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combi_graph, mpqp_hip_combinatorial
wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
cap = int(sys.argv[2]) if len(sys.argv) > 2 else None
variant = sys.argv[3] if len(sys.argv) > 3 else 'combinatorial_graph'
run = mpqp_hip_combi_graph.solve if variant == 'combinatorial_graph' else mpqp_hip_combi_graph.solve_graph
prog = bench.build_program(wl)
for rep in range(2):
    prof = []
    t = time.perf_counter()
    sol = run(prog, profile=prof, max_candidates=cap)
    dt = time.perf_counter() - t
    n = sum(p['candidates'] for p in prof)
    print(f'{wl} {variant}: {len(sol.critical_regions)} regions, {n} active sets examined in {len(prof)} waves, {dt * 1e3:.1f} ms '
          f'({n / dt:.3g} sets/s, {len(sol.critical_regions) / dt:.3g} regions/s)')
print('waves:', [(p['candidates'], p['regions']) for p in prof][:40])
ml = bench.WORKLOADS[wl][2]
t = time.perf_counter(); ref = mpqp_hip_combinatorial.solve(prog, max_levels=ml); dt = time.perf_counter() - t
keys = {tuple(r.active_set) for r in sol.critical_regions}
rk = {tuple(r.active_set) for r in ref.critical_regions}
print(f'combinatorial levels 1-{ml}: {len(rk)} regions in {dt * 1e3:.1f} ms; contained in the graph solution: {len(rk & keys)}')

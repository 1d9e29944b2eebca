"""The geometric algorithm on a bench workload (run on the GPU box): python tools/geometric_run.py [c2|c3|c4] [max_regions]"""
import sys, time
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_geometric, mpqp_hip_combi_graph
wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
cap = int(sys.argv[2]) if len(sys.argv) > 2 else None
prog = bench.build_program(wl)
prof = []
t = time.perf_counter(); sol = mpqp_hip_geometric.solve(prog, profile=prof, max_regions=cap); dt = time.perf_counter() - t
print(f'{wl} geometric: {len(sol.critical_regions)} regions in {dt:.2f} s ({len(sol.critical_regions) / dt:.3g} regions/s), rounds {len(prof)}, '
      f'facet LPs {sum(p["facets"] for p in prof)}, QPs {sum(p["qps"] for p in prof)}, complete {sol.is_complete}')
print([(p['regions_in'], p['regions_out']) for p in prof][:40])
g = mpqp_hip_combi_graph.solve_graph(prog)
a = {tuple(r.active_set) for r in sol.critical_regions}; b = {tuple(r.active_set) for r in g.critical_regions}
print('graph:', len(b), 'regions; geometric only', len(a - b), 'graph only', len(b - a))

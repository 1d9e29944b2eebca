"""Where the time of a connected-graph solve goes (run on the GPU box): python tools/graph_profile.py [c4|c3] [graph|combinatorial_graph]"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combi_graph
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
variant = sys.argv[2] if len(sys.argv) > 2 else 'graph'
run = mpqp_hip_combi_graph.solve if variant == 'combinatorial_graph' else mpqp_hip_combi_graph.solve_graph
prog = bench.build_program(wl)
run(prog, max_candidates=200000)
pr = cProfile.Profile(); pr.enable(); t = time.perf_counter(); prof = []; sol = run(prog, profile=prof); dt = time.perf_counter() - t; pr.disable()
print(f'{wl} {variant}: {len(sol.critical_regions)} regions, {sum(p["candidates"] for p in prof)} sets, {len(prof)} waves, {dt:.3f} s')
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)

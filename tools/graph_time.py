"""Steady-state time of the complete solution by the graph traversal (run on the GPU box): python tools/graph_time.py [c4] [reps]"""
import sys, time, gc
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combi_graph
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
prog = bench.build_program(wl)
for rep in range(reps):
    gc.collect()
    t = time.perf_counter()
    sol = mpqp_hip_combi_graph.solve_graph(prog)
    dt = time.perf_counter() - t
    print(f'{wl} graph run {rep}: {len(sol.critical_regions)} regions, {dt * 1e3:.1f} ms')
    del sol

"""mpLPs: the geometric algorithm (probe LPs as a device batch) against the combinatorial one (run on the GPU box)."""
import sys, time, warnings
sys.path.insert(0, '.')
from ppopt_amd.problem_generator import generate_mplp
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial, mpqp_hip_geometric
for (nx, nt, m, seed) in ((6, 3, 14, 1), (8, 4, 16, 2), (10, 4, 20, 3), (12, 5, 20, 4)):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = generate_mplp(nx, nt, m, seed)
    mpqp_hip_geometric.solve(prog)
    t = time.perf_counter(); g = mpqp_hip_geometric.solve(prog); tg = time.perf_counter() - t
    mpqp_hip_combinatorial.solve(prog)
    prof = []
    t = time.perf_counter(); c = mpqp_hip_combinatorial.solve(prog, profile=prof); tc = time.perf_counter() - t
    kg, kc = {tuple(r.active_set) for r in g.critical_regions}, {tuple(r.active_set) for r in c.critical_regions}
    print(f'mpLP ({nx},{nt},{m},{seed}) n_c={prog.num_constraints()}: combinatorial {len(kc)} regions, {sum(p["candidates"] for p in prof)} candidates, {tc*1e3:.1f} ms | '
          f'geometric {len(kg)} regions in {tg*1e3:.1f} ms | geometric subset {kg <= kc}, missing {len(kc - kg)}')

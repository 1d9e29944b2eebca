"""Completeness of a connected-graph solution, without an oracle (run on the GPU box):
    python tools/graph_verify.py [c4|c3] [graph|combinatorial_graph] [points]
Random parameter points of the box are located in the solution (device locator).  A point that lies in no region must be a
point where the program has no feasible x at all -- checked as one device LP batch {x : A x <= b + F theta}; and at the located
points the region's law must satisfy the KKT conditions (Solution.kkt_residuals on a subsample)."""
import sys, time
sys.path.insert(0, '.')
import numpy
import bench
from ppopt_amd import _lib
from ppopt_amd.mp_solvers import mpqp_hip_combi_graph
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
variant = sys.argv[2] if len(sys.argv) > 2 else 'graph'
m = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
run = mpqp_hip_combi_graph.solve if variant == 'combinatorial_graph' else mpqp_hip_combi_graph.solve_graph
prog = bench.build_program(wl)
t = time.perf_counter(); sol = run(prog); dt = time.perf_counter() - t
print(f'{wl} {variant}: {len(sol.critical_regions)} regions in {dt:.2f} s')
nt = prog.num_t()
# bounding box of the parameter set from its rows (the bench programs use boxes)
lo, hi = numpy.full(nt, -numpy.inf), numpy.full(nt, numpy.inf)
for row, rhs in zip(prog.A_t, prog.b_t.ravel()):
    nz = numpy.flatnonzero(numpy.abs(row) > 1e-12)
    if len(nz) == 1:
        j = nz[0]
        if row[j] > 0: hi[j] = min(hi[j], rhs / row[j])
        else: lo[j] = max(lo[j], rhs / row[j])
rng = numpy.random.default_rng(1)
th = lo + rng.random((m, nt)) * (hi - lo)
t = time.perf_counter(); x, idx = sol.evaluate_batch(th); dt = time.perf_counter() - t
inside = idx >= 0
print(f'{m} points located in {dt * 1e3:.1f} ms: {int(inside.sum())} inside a region, {int((~inside).sum())} in none')
out = numpy.flatnonzero(~inside)
if len(out):
    # is there any x with A x <= b + F theta (equalities as posed)?  one LP per point, shared A
    bb = prog.b.ravel()[None, :] + th[out] @ prog.F.T
    flags = numpy.zeros((len(out), prog.A.shape[0]), dtype=numpy.uint8)
    flags[:, list(prog.equality_indices)] = 1
    st, _, _, _ = _lib.lp_solve_batch(prog.A, bb, None, flags, want_x=False)
    feas = st == _lib.LP_OPTIMAL
    print(f'of the {len(out)} points outside every region: {int(feas.sum())} have a feasible x (MISSING REGIONS if > 0; points on region boundaries excepted)')
    if feas.any():
        # how far outside? distance to the nearest region boundary is not available cheaply: report the relaxed locate
        sol.point_location_tolerance = 1e-6
        sol._locator_key = None
        again = sol.get_region_batch(th[out][feas])
        print('   located with tolerance 1e-6:', int((again >= 0).sum()), 'of', int(feas.sum()))
chk = numpy.flatnonzero(inside)[:300]
worst = 0.0
for p in chk:
    r = sol.kkt_residuals(sol.critical_regions[int(idx[p])], th[p].reshape(-1, 1))
    worst = max(worst, max(v for v in r.values() if isinstance(v, float)))
print(f'KKT residuals at {len(chk)} located points: worst {worst:.2e}')

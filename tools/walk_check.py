"""Walk locator against the list scan on a complete graph solution (run on the GPU box): python tools/walk_check.py [c4|c3] [points]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combi_graph, mpqp_hip_combinatorial
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
m = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
prog = bench.build_program(wl)
sol = mpqp_hip_combi_graph.solve_graph(prog)
print(len(sol.critical_regions), 'regions')
from test_gpu_parity import _theta_samples
th = _theta_samples(prog, m, 9)
t = time.perf_counter(); loc = sol.locator(); print('locator build %.1f ms, adjacency %s' % ((time.perf_counter() - t) * 1e3, loc.has_adjacency))
for rep in range(2):
    t = time.perf_counter(); xw, iw = sol.evaluate_batch(th); tw = time.perf_counter() - t
print('walk: %d points in %.2f ms (kernel %.2f ms) = %.3g points/s' % (m, tw * 1e3, loc.last_ms, m / tw))
big = _theta_samples(prog, 1000000, 10)
t = time.perf_counter(); xb, ib = sol.evaluate_batch(big); tb = time.perf_counter() - t
print('walk: 1e6 points in %.1f ms (kernel %.1f ms) = %.3g points/s, located %d' % (tb * 1e3, loc.last_ms, 1e6 / tb, (ib >= 0).sum()))
sol.use_walk = False
sub = th[:min(m, 20000)]
t = time.perf_counter(); xs, is_ = sol.evaluate_batch(sub); ts = time.perf_counter() - t
print('scan: %d points in %.1f ms = %.3g points/s' % (len(sub), ts * 1e3, len(sub) / ts))
same = iw[:len(sub)] == is_
print('same region index: %d of %d; located %d' % (same.sum(), len(sub), (is_ >= 0).sum()))
bad = numpy.flatnonzero(~same)
for p in bad[:10]:
    print('  point', p, 'walk', iw[p], 'scan', is_[p], 'max |x diff|', numpy.nanmax(numpy.abs(xw[p] - xs[p])) if iw[p] >= 0 and is_[p] >= 0 else None)

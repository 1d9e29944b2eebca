"""Debug build only (-DMPC_XQ_HIST, MPC_LIB_PATH=tools/ubench/libmpc_xqhist.so): after how many ratio tests k_xq decides a candidate of the
last level.  python tools/xq_hist.py [workload]"""
import ctypes, os, sys
sys.path.insert(0, '.')
import numpy
import bench
from ppopt_amd import _lib
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
prog = bench.build_program(wl)
ml = bench.WORKLOADS[wl][2]
L = _lib.load()
buf = (ctypes.c_ulonglong * 128)()
m.solve(prog, max_levels=ml)
L.mpc_debug_xq_hist(buf)          # reset
prof = []
m.solve(prog, max_levels=ml, profile=prof)
assert L.mpc_debug_xq_hist(buf) == 0
allv = numpy.array(list(buf))
h = allv[:60].reshape(3, 20)
h2 = allv[64:104].reshape(2, 20)
tot = h.sum()
print(wl, 'k_xq candidates', tot, '(levels:', [(p['depth'], p['candidates']) for p in prof], ')')
for name, row in zip(('undecided', 'infeasible', 'feasible'), h):
    print('%-10s' % name, ' '.join('%7d' % v for v in row), ' sum %d (%.1f %%)' % (row.sum(), 100.0 * row.sum() / max(tot, 1)))
cum = numpy.cumsum(h[1] + h[2])
print('decided within n ratio tests (%):', ' '.join('%.1f' % (100.0 * c / max(tot, 1)) for c in cum))
print('candidates by "some column lets the new row leave at once" (rows: no / yes) x ratio tests k_xq needed:')
for name, row in zip(('no column', 'a column'), h2):
    print('%-10s' % name, ' '.join('%7d' % v for v in row), ' sum %d (%.1f %%)' % (row.sum(), 100.0 * row.sum() / max(h2.sum(), 1)))

import sys, gc, weakref, collections
sys.path.insert(0, '.')
import numpy
import bench
from ppopt_amd import _lib
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
prog = bench.build_program('c2')
sol = mpqp_hip_combinatorial.solve(prog)
sol = None
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
sol = mpqp_hip_combinatorial.solve(prog)
sol = None
n = gc.collect()
print('collected', n)
print(collections.Counter(type(o).__name__ for o in gc.garbage).most_common(12))
for o in gc.garbage:
    if isinstance(o, dict) and len(o) < 8:
        print('dict keys', list(o.keys())[:8])
        break
for o in gc.garbage:
    if type(o).__name__ in ('cell', 'function', 'tuple'):
        print(type(o).__name__, repr(o)[:200])

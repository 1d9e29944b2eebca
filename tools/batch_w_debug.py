"""Debug: region records of a batch member against its single-program run, field by field (run on the GPU box)."""
import os, sys
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_golden
from test_gpu_parity import engine_from_golden
from test_gpu_batch import MIXED, _levels_alone, _snapshot
from ppopt_amd import _lib
goldens = [load_golden(n) for n in MIXED]
n_levels = [None if bool(g['complete']) else 3 for g in goldens]
alone = [_levels_alone(g, nl) for g, nl in zip(goldens, n_levels)]
engs = [engine_from_golden(g) for g in goldens]
depth_max = [max(e.n_x, e.n_t) - e.n_eq if nl is None else min(max(e.n_x, e.n_t) - e.n_eq, nl) for e, nl in zip(engs, n_levels)]
for e in engs:
    e.pruned_clear(); e.frontier_root()
active = list(range(len(engs))); depth = 0; ndiff = ntot = 0
while active:
    gens = [depth + 1 != depth_max[i] for i in active]
    stats, n_shared = _lib.Engine.level_run_batch([engs[i] for i in active], gens)
    nxt = []
    for i, st, gen in zip(active, stats, gens):
        a, b = _snapshot(engs[i], st, gen), alone[i][depth]
        assert numpy.array_equal(a['status'], b['status']), (MIXED[i], depth)
        for c in a['regs']:
            ntot += 1
            if a['regs'][c] != b['regs'][c]:
                ndiff += 1
                ha, hb = numpy.frombuffer(a['regs'][c][0], dtype=numpy.int32), numpy.frombuffer(b['regs'][c][0], dtype=numpy.int32)
                da, db = numpy.frombuffer(a['regs'][c][1]), numpy.frombuffer(b['regs'][c][1])
                ea, eb = numpy.frombuffer(a['regs'][c][2]), numpy.frombuffer(b['regs'][c][2])
                print(MIXED[i], 'depth', depth, 'cand', c, 'int head differs at', numpy.nonzero(ha != hb)[0].tolist()[:12], 'nE', ha[2], hb[2],
                      'head_d max diff', float(numpy.abs(da - db).max()) if da.shape == db.shape else 'shape',
                      'rows max diff', float(numpy.abs(ea - eb).max()) if ea.shape == eb.shape else ('shape', ea.shape, eb.shape))
                if ha[2] != hb[2] or True:
                    print('   batch ints', ha.tolist()); print('   alone ints', hb.tolist())
        if gen and st.n_children:
            engs[i].frontier_advance(); nxt.append(i)
    active = nxt; depth += 1
print('regions', ntot, 'differing', ndiff)

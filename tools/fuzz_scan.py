"""Counts verdict differences between the GPU path and the CPU oracle over many small random programs (GPU box)."""
import sys, warnings
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy
from ppopt_amd import MPQP_Program, Solver, problem_generator as pg
from oracle import oracle as orc
orc.build()
n_prob = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = numpy.random.default_rng(99)
tot = diff = 0
kinds = {}
for it in range(n_prob):
    nx, nt = int(rng.integers(3, 8)), int(rng.integers(2, 6))
    m = int(rng.integers(nx + 3, 3 * nx + 4))
    seed = 1000 + it
    d = pg.generate_mpqp_data(nx, nt, m, seed)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], solver=Solver())
    P = orc.OracleProblem(prog.A, prog.b, prog.F, prog.c, prog.H, prog.Q, prog.A_t, prog.b_t, len(prog.equality_indices))
    eng = prog.engine(0)
    eng.pruned_clear(); eng.frontier_root()
    depth = 0
    max_depth = max(nx, nt)
    while True:
        depth += 1
        gen = depth != max_depth
        st = eng.level_run(gen)
        gc, gs = eng.frontier_get(), eng.level_status()
        if len(gc) > 4000:
            break
        ost, _ = P.check_level(gc, 0, False)
        tot += len(gc)
        for c, v, ov in zip(gc.tolist(), gs.tolist(), ost.tolist()):
            if v != ov:
                diff += 1
                kinds[(ov, v)] = kinds.get((ov, v), 0) + 1
                print('CASE', nx, nt, m, seed, c, 'oracle', ov, 'gpu', v, flush=True)
        if not gen or st.n_children == 0:
            break
        eng.frontier_advance()
    prog._engine = None if hasattr(prog, '_engine') else None
print('candidates', tot, 'differ', diff, 'by (oracle, gpu):', kinds)

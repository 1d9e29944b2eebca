"""Counts verdict / region differences between the GPU path and the CPU oracle over many small programs (GPU box).

    python tools/fuzz_scan.py <n_programs> [mpqp|mpqp_eq|mplp|mpc|big|open] [rng seed]
"""
import sys, warnings
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import numpy
from ppopt_amd import MPLP_Program, MPQP_Program, Solver, problem_generator as pg
from ppopt_amd.region_batch import RegionBatch
from oracle import oracle as orc
from conftest import kkt_condition, rel_err
from _fuzz_draw import draw
orc.build()
n_prob = int(sys.argv[1]) if len(sys.argv) > 1 else 60
kind = sys.argv[2] if len(sys.argv) > 2 else 'mpqp'
rng = numpy.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 99)
tot = diff = nreg = regdiff = coefbad = 0
kinds = {}
for it in range(n_prob):
    print('program', it, flush=True) if kind == 'big' else None
    d, tag = draw(kind, rng, pg)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if kind == 'mplp':
            prog = MPLP_Program(d['A'], d['b'], d['c'], d['H'], d['A_t'], d['b_t'], d['F'], equality_indices=d.get('equality_indices'), solver=Solver())
        else:
            prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=d.get('equality_indices'), solver=Solver())
    Q = None if kind == 'mplp' else prog.Q
    P = orc.OracleProblem(prog.A, prog.b, prog.F, prog.c, prog.H, Q, prog.A_t, prog.b_t, len(prog.equality_indices))
    if prog.num_constraints() > 128:      # outside the device path (size limits)
        print('skipped', tag, 'n_c', prog.num_constraints(), flush=True)
        continue
    eng = prog.engine(0)
    eng.pruned_clear(); eng.frontier_root()
    depth = 0
    max_depth = max(eng.n_x, eng.n_t) - eng.n_eq
    while max_depth > 0:
        depth += 1
        gen = depth != max_depth
        st = eng.level_run(gen)
        gc, gs = eng.frontier_get(), eng.level_status()
        if len(gc) == 0 or (kind != 'big' and len(gc) > 6000) or len(gc) > 400000:
            break
        if len(gc) > 4000:   # big levels: an evenly strided sample plus some of the regions
            pick = numpy.unique(numpy.concatenate([numpy.linspace(0, len(gc) - 1, 3000).astype(numpy.int64), numpy.flatnonzero(gs == 3)[:300]]))
            gc, gs = numpy.ascontiguousarray(gc[pick]), gs[pick]
        ost, oregs = P.check_level(gc, 0, True)
        tot += len(gc)
        for c, v, ov in zip(gc.tolist(), gs.tolist(), ost.tolist()):
            if v != ov:
                diff += 1
                kinds[(ov, v)] = kinds.get((ov, v), 0) + 1
                print('CASE', kind, tag, c, 'oracle', ov, 'gpu', v, 'cond %.1e' % kkt_condition(P, c), flush=True)
        if st.n_regions:
            hd, hi, er, kk, slots = eng.level_regions_slots()
            mine = {tuple(r.active_set): r for r in RegionBatch(hd, hi, er, eng.n_x, eng.n_t, eng.n_c, eng.n_tc, kk, slots).regions()}
            for j, q in oregs.items():
                key = tuple(q['active_set'])
                if key not in mine:
                    continue
                nreg += 1
                r = mine[key]
                tol = max(1e-8, 4e-16 * kkt_condition(P, list(key)))
                if any(rel_err(getattr(r, f), q[f]) > tol for f in ('A', 'b', 'C', 'd')):
                    coefbad += 1
                    print('COEF', kind, tag, key, [float(rel_err(getattr(r, f), q[f])) for f in ('A', 'b', 'C', 'd')], flush=True)
                if not (r.omega_set == q['omega_set'] and r.lambda_set == q['lambda_set'] and r.regular_set == q['regular_set']):
                    regdiff += 1
                    print('FACETS', kind, tag, key, 'cond %.1e' % kkt_condition(P, list(key)), flush=True)
        if not gen or st.n_children == 0:
            break
        eng.frontier_advance()
    eng.close()
    prog._engine = None
print(kind, 'programs', n_prob, 'candidates', tot, 'verdicts differ', diff, kinds, '| regions compared', nreg, 'facet sets differ', regdiff, 'coefficients off', coefbad)

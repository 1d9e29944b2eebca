"""Fuzz of the shared-launch path: random mpQPs / mpLPs of many shapes solved together (solve_many) against solved one by one (solve):
regions must be the same sets with bit-identical numbers.  usage: python tools/fuzz_batch.py [n_programs] [seed] [batch] [max_levels]"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy  # noqa: E402

from ppopt_amd import MPLP_Program, MPQP_Program, problem_generator as pg  # noqa: E402
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial  # noqa: E402

args = [int(v) for v in sys.argv[1:]]
n_prog, seed, batch, max_levels = (args + [400, 2026, 50, 5][len(args):])[:4]
warnings.simplefilter('ignore')


def make(j):
    rng = numpy.random.default_rng(seed * 100003 + j)
    nx, nt, m = int(rng.integers(2, 17)), int(rng.integers(1, 11)), int(rng.integers(4, 26))
    d = pg.generate_mpqp_data(nx, nt, m, seed * 7919 + j)
    if j % 5 == 4:
        d['equality_indices'] = [0]
    if j % 7 == 6:
        return MPLP_Program(d['A'], d['b'], d['c'], d['H'], d['A_t'], d['b_t'], d['F'], equality_indices=d['equality_indices'])
    return MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=d['equality_indices'])


tot = {'programs': 0, 'regions': 0, 'candidates': 0, 'shared': 0, 'members': 0, 'mismatch': 0, 'skipped': 0}
only = [int(v) for v in os.environ.get('FUZZ_ONLY', '').split(',') if v]
for lo in range(0, n_prog, batch):
    idx, one, progs = [], [], []
    for j in range(lo, min(n_prog, lo + batch)):
        if only and j not in only:
            continue
        try:
            p1, p2 = make(j), make(j)
        except Exception:
            tot['skipped'] += 1
            continue
        try:
            sol1 = mpqp_hip_combinatorial.solve(p1, max_levels=max_levels)
        except Exception as ex:
            print('SOLVE FAILED program', j, 'n_x', p1.num_x(), 'n_t', p1.num_t(), 'n_c', p1.num_constraints(), 'n_eq', len(p1.equality_indices),
                  'A_t', p1.A_t.shape, type(p1).__name__, str(ex)[:160], flush=True)
            tot['skipped'] += 1
            p1.release_engine()
            continue
        idx.append(j)
        one.append(sol1)
        p1.release_engine()
        progs.append(p2)
    prof = []
    many = mpqp_hip_combinatorial.solve_many(progs, max_levels=max_levels, profile=prof)
    for p in progs:
        p.release_engine()
    tot['shared'] += sum(p['shared_launches'] for p in prof)
    tot['members'] += sum(p['members'] for p in prof)
    tot['candidates'] += sum(p['candidates'] for p in prof)
    for j, a, b in zip(idx, one, many):
        tot['programs'] += 1
        tot['regions'] += len(a.critical_regions)
        ka = {tuple(r.active_set): r for r in a.critical_regions}
        kb = {tuple(r.active_set): r for r in b.critical_regions}
        bad = ka.keys() != kb.keys()
        n_idx = n_num = 0
        worst = 0.0
        levels = set()
        if not bad:
            for key, r1 in ka.items():
                r2 = kb[key]
                if r1.omega_set != r2.omega_set or r1.lambda_set != r2.lambda_set or r1.regular_set != r2.regular_set:
                    n_idx += 1
                    levels.add(len(key))
                    continue
                for fld in ('A', 'b', 'C', 'd', 'E', 'f'):
                    x1, x2 = numpy.asarray(getattr(r1, fld)), numpy.asarray(getattr(r2, fld))
                    if x1.shape != x2.shape:
                        n_idx += 1
                        levels.add(len(key))
                        break
                    if x1.tobytes() != x2.tobytes():
                        n_num += 1
                        levels.add(len(key))
                        worst = max(worst, float(numpy.max(numpy.abs(x1 - x2) / (1.0 + numpy.abs(x2)))))
                        break
            bad = bool(n_idx or n_num)
        if bad:
            tot['mismatch'] += 1
            print('MISMATCH program', j, 'regions', len(ka), len(kb), 'same active sets', ka.keys() == kb.keys(), 'regions with other index sets / row counts', n_idx,
                  'with other numbers', n_num, 'worst relative difference', worst, 'cardinalities', sorted(levels), flush=True)
    print(f'batch {lo // batch}: {tot}', flush=True)
print('RESULT', tot)

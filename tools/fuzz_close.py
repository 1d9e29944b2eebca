"""Programs whose parameter set has no vertex after the presolve: solved with the closing rows of MPLP_Program._engine_parameter_rows
(register-resident kernels) against without (MPC_NO_THETA_CLOSE=1, LDS-engine kernels): same region sets, index sets, coefficients 1e-8;
disputed regions are put to the CPU oracle.  usage: python tools/fuzz_close.py [n_programs_to_scan] [seed]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy  # noqa: E402

from ppopt_amd import MPQP_Program, problem_generator as pg  # noqa: E402
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial  # noqa: E402

args = [int(v) for v in sys.argv[1:]]
n_scan, seed = (args + [400, 7][len(args):])[:2]
warnings.simplefilter('ignore')


def programs():
    for j in range(n_scan):
        rng = numpy.random.default_rng(seed * 1000003 + j)
        kind = j % 3
        if kind == 0:
            d = pg.double_integrator_data(int(rng.integers(2, 7)), float(rng.uniform(6, 40)), float(rng.uniform(0.5, 2)))
        else:
            nx, nt, m = int(rng.integers(3, 13)), int(rng.integers(2, 7)), int(rng.integers(6, 20))
            d = pg.generate_mpqp_data(nx, nt, m, seed * 7919 + j)
            if kind == 2:      # drop some parameter rows: the set loses its vertex
                keep = rng.random(d['A_t'].shape[0]) < 0.5
                d['A_t'], d['b_t'] = d['A_t'][keep], d['b_t'][keep]
                if d['A_t'].shape[0] == 0:
                    continue
        yield j, d


tot = {'scanned': 0, 'without_vertex': 0, 'closed': 0, 'regions': 0, 'mismatch': 0, 'failed': 0}
t_close = t_plain = 0.0
for j, d in programs():
    tot['scanned'] += 1
    try:
        def build():
            return MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=d['equality_indices'])
        p1 = build()
    except Exception:
        continue
    if p1.A_t.shape[0] >= p1.num_t() and numpy.linalg.matrix_rank(p1.A_t) >= p1.num_t():
        continue
    tot['without_vertex'] += 1
    try:
        eng = p1.engine(0, closed=True)
        if eng.n_tc == p1.A_t.shape[0]:
            p1.release_engine()
            continue          # unbounded in theta, or a box of big-M size: no closing rows
        tot['closed'] += 1
        t0 = time.perf_counter(); a = mpqp_hip_combinatorial.solve(p1, max_levels=6); t_close += time.perf_counter() - t0
        p1.release_engine()
        os.environ['MPC_NO_THETA_CLOSE'] = '1'
        p2 = build()
        t0 = time.perf_counter(); b = mpqp_hip_combinatorial.solve(p2, max_levels=6); t_plain += time.perf_counter() - t0
        p2.release_engine()
        del os.environ['MPC_NO_THETA_CLOSE']
    except Exception as ex:
        os.environ.pop('MPC_NO_THETA_CLOSE', None)
        tot['failed'] += 1
        print('FAILED', j, str(ex)[:150], flush=True)
        continue
    ka = {tuple(r.active_set): r for r in a.critical_regions}
    kb = {tuple(r.active_set): r for r in b.critical_regions}
    tot['regions'] += len(kb)
    bad = None
    if ka.keys() != kb.keys():
        bad = f'region sets differ: {len(ka)} / {len(kb)}'
    else:
        for key, r1 in ka.items():
            r2 = kb[key]
            if r1.omega_set != r2.omega_set or r1.lambda_set != r2.lambda_set or r1.regular_set != r2.regular_set:
                bad = f'index sets differ at {key}'
                break
            for fld in ('A', 'b', 'C', 'd', 'E', 'f'):
                x1, x2 = numpy.asarray(getattr(r1, fld)), numpy.asarray(getattr(r2, fld))
                if x1.shape != x2.shape or numpy.max(numpy.abs(x1 - x2) / (1 + numpy.abs(x2)), initial=0.0) > 1e-8:
                    bad = f'{fld} differs at {key}'
                    break
            if bad:
                break
    if bad and tot['mismatch'] < 10:
        from oracle import oracle as orc
        orc.build()
        P = orc.OracleProblem(p2.A, p2.b, p2.F, p2.c, p2.H, p2.Q, p2.A_t, p2.b_t, len(p2.equality_indices))
        _, oregs, _ = P.solve(threads=0, max_levels=6)
        ko = {tuple(r['active_set']): r for r in oregs}
        agree = {'closed': 0, 'plain': 0, 'both': 0, 'neither': 0}
        for key in ka:
            if key not in ko or key not in kb:
                continue
            q = ko[key]
            ok1 = ka[key].omega_set == q['omega_set'] and ka[key].lambda_set == q['lambda_set'] and ka[key].regular_set == q['regular_set']
            ok2 = kb[key].omega_set == q['omega_set'] and kb[key].lambda_set == q['lambda_set'] and kb[key].regular_set == q['regular_set']
            agree['both' if ok1 and ok2 else ('closed' if ok1 else ('plain' if ok2 else 'neither'))] += 1
        bad += f' | oracle: {len(ko)} regions, index sets agree with {agree}'
    if bad:
        tot['mismatch'] += 1
        print('MISMATCH', j, 'n_x', p1.num_x(), 'n_t', p1.num_t(), 'n_c', p1.num_constraints(), 'A_t', p1.A_t.shape, bad, flush=True)
print('RESULT', tot, f'solve time with closing rows {1e3 * t_close:.1f} ms, without {1e3 * t_plain:.1f} ms')

"""Golden-vector generator: runs the REAL reference (imported through ref_shims) on the
configuration problems and writes small fixtures to tests/golden/.

Runs only in the build container (needs /root/reference).  Usage:
    python oracle/ref_harness/gen_goldens.py [name ...]

What is captured per problem:
  raw_*   raw constructor inputs (from ppopt_amd.problem_generator -- own code)
  proc_*  the reference's presolved A,b,F,A_t,b_t,equality_indices (mplp_program.py:60-134,285-306)
  L{i}_cands / L{i}_verdict   per BFS level of the parallel driver
          (mpqp_parrallel_combinatorial.py:67-150, with `shuffle` replaced by sorted order):
          the candidate active sets and the verdict of the reference's own primitives
          0 infeasible/rank-deficient  1 feasible, not optimal  2 optimal, region is None
          3 region                     4 optimal, KKT solve raised LinAlgError (mpqp_program.py:187)
  R_*     every CriticalRegion field (critical_region.py:34-48), regions sorted by active set
  base_verdict  verdict of the base active set (= equality_indices), driver lines 142-146
The verdicts come from calling program.check_feasibility / check_optimality /
gen_cr_from_active_set exactly in the order full_process does (lines 17-64); the harness
asserts that the reference's own full_process and solve() agree with the trace.
"""
import os
import sys
import time
import warnings

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shims  # noqa: E402

ppopt = ref_shims.load_reference()

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location('pg', os.path.join(ROOT, 'ppopt_amd', 'problem_generator.py'))
pg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(pg)

from ppopt.mp_solvers import mpqp_combinatorial, mpqp_parrallel_combinatorial  # noqa: E402
from ppopt.mp_solvers.solver_utils import CombinationTester, generate_children_sets  # noqa: E402
from ppopt.mplp_program import MPLP_Program  # noqa: E402
from ppopt.mpqp_program import MPQP_Program  # noqa: E402
from ppopt.utils.mpqp_utils import gen_cr_from_active_set  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def build_reference_program(d, post_process=True):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if d['Q'] is None:
            return MPLP_Program(d['A'], d['b'], d['c'], d['H'], d['A_t'], d['b_t'], d['F'],
                                equality_indices=list(d['equality_indices']), post_process=post_process)
        return MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'],
                            equality_indices=list(d['equality_indices']), post_process=post_process)


def classify(program, active_set):
    """The full_process state machine (mpqp_parrallel_combinatorial.py:17-64) on the reference's
    own primitives, returning (verdict, region, cond(KKT) or nan)."""
    if not program.check_feasibility(active_set):
        return 0, None, numpy.nan
    if not program.check_optimality(active_set):
        return 1, None, numpy.nan
    cond = numpy.nan
    if hasattr(program, 'Q'):
        A_hat = program.A[active_set]
        k = len(active_set)
        M = numpy.block([[A_hat, numpy.zeros((k, k))], [program.Q, A_hat.T]])
        cond = numpy.linalg.cond(M)
    try:
        region = gen_cr_from_active_set(program, active_set)
    except numpy.linalg.LinAlgError:
        return 4, None, cond
    if region is None:
        return 2, None, cond
    return 3, region, cond


def _worker(args):
    program, cand = args
    return classify(program, cand)


def trace(program, max_levels=None, pool=None, check_full_process=False):
    murder = CombinationTester()
    e = len(program.equality_indices)
    n_c = program.num_constraints()
    max_depth = max(program.num_x(), program.num_t()) - e
    to_check = generate_children_sets(program.equality_indices, n_c)
    levels = []
    regions = []
    is_mplp = type(program) is MPLP_Program
    for i in range(max_depth):
        if max_levels is not None and i >= max_levels:
            break
        gen_children = i + 1 != max_depth
        to_check = sorted(to_check)
        t0 = time.time()
        if pool is not None and len(to_check) > 64:
            out = pool.map(_worker, [(program, c) for c in to_check])
        else:
            out = [classify(program, c) for c in to_check]
        verdicts = numpy.array([o[0] for o in out], dtype=numpy.uint8)
        conds = numpy.array([o[2] for o in out], dtype=numpy.float64)
        print(f'  level {i + 1}: {len(to_check)} candidates, hist {numpy.bincount(verdicts, minlength=5)}, '
              f'{time.time() - t0:.1f}s', flush=True)
        levels.append((numpy.array(to_check, dtype=numpy.int32).reshape(len(to_check), e + i + 1), verdicts, conds))
        pruned = set()
        future = []
        for cand, (v, reg, _) in zip(to_check, out):
            if check_full_process:
                ref = mpqp_parrallel_combinatorial.full_process(program, cand, murder, gen_children)
                assert (ref[0] is None) == (reg is None)
                assert ref[1] == ({tuple(cand)} if v in (0, 2) else set()), (cand, v, ref[1])
            if v in (0, 2, 4):
                pruned.add(tuple(cand))
                continue
            if v == 3:
                regions.append(reg)
            if gen_children:
                kids = generate_children_sets(cand, n_c, murder)
                if v == 1 and is_mplp:
                    kids = [k for k in kids if not (k[-1] >= len(k) + n_c - program.num_x())]
                future.extend(kids)
        if not gen_children:
            break
        murder.add_combos(pruned)
        to_check = future
        if len(to_check) == 0:
            break
    base = classify(program, list(program.equality_indices))
    if base[0] == 3:
        regions.append(base[1])
    return levels, regions, base[0]


def pad_int(lists, width=None):
    width = max([len(x) for x in lists] + [1]) if width is None else width
    out = -numpy.ones((len(lists), width), dtype=numpy.int32)
    for i, x in enumerate(lists):
        out[i, :len(x)] = x
    return out


def pack_regions(regions, n_x, n_t):
    regions = sorted(regions, key=lambda r: (len(r.active_set), list(r.active_set)))
    nr = len(regions)
    kmax = max([len(r.active_set) for r in regions] + [1])
    emax = max([r.E.shape[0] for r in regions] + [1])
    out = {
        'R_k': numpy.array([len(r.active_set) for r in regions], dtype=numpy.int32),
        'R_active': pad_int([r.active_set for r in regions], kmax),
        'R_A': numpy.zeros((nr, n_x, n_t)), 'R_b': numpy.zeros((nr, n_x)),
        'R_C': numpy.zeros((nr, kmax, n_t)), 'R_d': numpy.zeros((nr, kmax)),
        'R_nE': numpy.array([r.E.shape[0] for r in regions], dtype=numpy.int32),
        'R_E': numpy.zeros((nr, emax, n_t)), 'R_f': numpy.zeros((nr, emax)),
        'R_omega': pad_int([r.omega_set for r in regions]),
        'R_lambda': pad_int([r.lambda_set for r in regions]),
        'R_regular_idx': pad_int([r.regular_set[0] for r in regions]),
        'R_regular_con': pad_int([r.regular_set[1] for r in regions]),
    }
    for i, r in enumerate(regions):
        k = len(r.active_set)
        out['R_A'][i] = r.A
        out['R_b'][i] = r.b.flatten()
        out['R_C'][i, :k] = r.C
        out['R_d'][i, :k] = r.d.flatten()
        ne = r.E.shape[0]
        out['R_E'][i, :ne] = r.E
        out['R_f'][i, :ne] = r.f.flatten()
    return out


def generate(name, d, max_levels=None, post_process=True, run_solvers=True, pool=None, check_full_process=False):
    print(f'== {name}', flush=True)
    t0 = time.time()
    program = build_reference_program(d, post_process)
    out = {}
    for key in ('A', 'b', 'c', 'H', 'Q', 'A_t', 'b_t', 'F'):
        if d[key] is not None:
            out['raw_' + key] = d[key]
    out['raw_eq'] = numpy.array(d['equality_indices'], dtype=numpy.int32)
    out['is_mplp'] = numpy.array(d['Q'] is None)
    out['post_process'] = numpy.array(post_process)
    for key in ('A', 'b', 'F', 'A_t', 'b_t'):
        out['proc_' + key] = getattr(program, key)
    out['proc_eq'] = numpy.array(program.equality_indices, dtype=numpy.int32)
    print(f'  presolved: n_x {program.num_x()} n_t {program.num_t()} n_c {program.num_constraints()} '
          f'e {len(program.equality_indices)} n_tc {program.A_t.shape[0]}', flush=True)
    levels, regions, base = trace(program, max_levels, pool, check_full_process)
    for i, (cands, verdicts, conds) in enumerate(levels):
        out[f'L{i}_cands'] = cands
        out[f'L{i}_verdict'] = verdicts
        out[f'L{i}_cond'] = conds
    out['n_levels'] = numpy.array(len(levels))
    out['complete'] = numpy.array(max_levels is None)
    out['base_verdict'] = numpy.array(base)
    out.update(pack_regions(regions, program.num_x(), program.num_t()))
    trace_sets = sorted(tuple(r.active_set) for r in regions)
    if run_solvers and max_levels is None:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            try:
                par = mpqp_parrallel_combinatorial.solve(program, 8)
                assert sorted(tuple(r.active_set) for r in par.critical_regions) == trace_sets
                ser = mpqp_combinatorial.solve(program)
                out['serial_region_count'] = numpy.array(len(ser.critical_regions))
                print(f'  reference solve(): parallel {len(par.critical_regions)} regions (== trace), '
                      f'serial {len(ser.critical_regions)} regions', flush=True)
            except numpy.linalg.LinAlgError as err:
                print(f'  reference solve() aborted: LinAlgError {err}', flush=True)
                out['reference_aborts'] = numpy.array(True)
    n_cand = sum(len(l[1]) for l in levels)
    out['ref_seconds'] = numpy.array(time.time() - t0)
    print(f'  {n_cand} candidates, {len(regions)} regions, {time.time() - t0:.1f}s', flush=True)
    numpy.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)


def lp_cases(seed=7, n=240):
    """Known-answer LPs at the deterministic-solver boundary (solver.py:211 -> cvxopt_interface.py:153):
    random feasible / infeasible / unbounded / equality-constrained problems with the reference's verdict."""
    from ppopt.solver import Solver
    rng = numpy.random.default_rng(seed)
    solver = Solver()
    recs = {}
    i = 0
    while i < n:
        nv = int(rng.integers(1, 9))
        m = int(rng.integers(nv + 1, nv + 14))
        A = numpy.round(rng.normal(size=(m, nv)) * 4) / 2
        kind = i % 4
        x0 = rng.normal(size=(nv, 1))
        b = A @ x0 + rng.random((m, 1)) * (2.0 if kind != 1 else 0.0)
        if kind == 1:  # make it infeasible-ish: contradictory pair
            A = numpy.vstack([A, -A[0:1]])
            b = numpy.vstack([b, -b[0:1] - 0.5 - rng.random()])
            m += 1
        neq = int(rng.integers(0, min(nv, 3) + 1)) if kind != 1 else 0
        eq = sorted(rng.choice(m, size=neq, replace=False).tolist())
        c = numpy.round(rng.normal(size=(nv, 1)) * 3) / 3 if kind != 3 else None
        if kind == 2:  # box to keep it bounded
            A = numpy.vstack([A, numpy.eye(nv), -numpy.eye(nv)])
            b = numpy.vstack([b, 10 * numpy.ones((2 * nv, 1))])
            m = A.shape[0]
        sol = solver.solve_lp(c, A, b, eq)
        recs[f'lp{i}_A'] = A
        recs[f'lp{i}_b'] = b
        recs[f'lp{i}_c'] = numpy.zeros((nv, 1)) if c is None else c
        recs[f'lp{i}_eq'] = numpy.array(eq, dtype=numpy.int32)
        recs[f'lp{i}_ok'] = numpy.array(sol is not None)
        recs[f'lp{i}_obj'] = numpy.array(float(numpy.asarray(sol.obj).reshape(-1)[0]) if sol is not None else numpy.nan)
        i += 1
    recs['n'] = numpy.array(n)
    numpy.savez_compressed(os.path.join(GOLDEN, 'lp_cases.npz'), **recs)
    oks = sum(bool(recs[f'lp{i}_ok']) for i in range(n))
    print(f'== lp_cases: {n} LPs, {oks} solved / {n - oks} None', flush=True)


REGISTRY = {
    # name: (data builder, kwargs for generate)
    'c1_transport_mplp': (lambda: pg.transport_mplp_data(), dict(check_full_process=True)),
    'transport_mpqp': (lambda: pg.transport_mpqp_data(), dict(check_full_process=True)),
    'dblint_n3': (lambda: pg.double_integrator_data(3), dict(check_full_process=True)),
    'c2_dblint_n5': (lambda: pg.double_integrator_data(5), {}),
    # BASELINE.json quotes config 2 with "~50 regions": the doc formulation's state box |x| <= 4 gives 9; with |x| <= 20 it is 52
    'c2_dblint_n5_x20': (lambda: pg.double_integrator_data(5, x_bound=20.0), {}),
    'rand_4_2_10_s0': (lambda: pg.generate_mpqp_data(4, 2, 10, 0), dict(check_full_process=True)),
    'rand_5_3_8_s3': (lambda: pg.generate_mpqp_data(5, 3, 8, 3), {}),
    'rand_6_3_12_s1': (lambda: pg.generate_mpqp_data(6, 3, 12, 1), {}),
    # random mpLPs (the reference's generate_mplp: the random mpQP without its Q): the dense-KKT / vertex-only branch
    'mplp_rand_4_2_10_s0': (lambda: {**pg.generate_mpqp_data(4, 2, 10, 0), 'Q': None}, dict(check_full_process=True)),
    'mplp_rand_5_3_12_s2': (lambda: {**pg.generate_mpqp_data(5, 3, 12, 2), 'Q': None}, {}),
    'quadtank_n2': (lambda: pg.quad_tank_data(2), {}),
    'quadtank_n3': (lambda: pg.quad_tank_data(3), {}),
    'c5_control_allocation': (lambda: pg.control_allocation_data(), {}),
    'c4_rand_20_8_20_s0': (lambda: pg.generate_mpqp_data(20, 8, 20, 0), dict(max_levels=2)),
    'c3_quadtank_n10': (lambda: pg.quad_tank_data(10), dict(max_levels=2)),
    # 82 rows after presolve (two tableau rows per lane on the device), big-M rows with right-hand sides of 1e7 left in: found by
    # tools/fuzz_scan.py big -- the reference and the device agree on [0, 2, 4] (a region), the CPU oracle's dense simplex does not
    'big_24_7_34_s430912': (lambda: pg.generate_mpqp_data(24, 7, 34, 430912), dict(max_levels=3)),
}

if __name__ == '__main__':
    import multiprocess

    os.makedirs(GOLDEN, exist_ok=True)
    names = sys.argv[1:] or (['lp_cases'] + list(REGISTRY))
    with multiprocess.Pool(8) as pool:
        for name in names:
            if name == 'lp_cases':
                lp_cases()
                continue
            builder, kw = REGISTRY[name]
            generate(name, builder(), pool=pool, **kw)

"""Several programs per launch (mpc_level_run_batch): a member's level must be what mpc_level_run computes for
that program alone -- candidates, statuses, region records, children and pruned masks IDENTICAL, bit for bit -- whatever else
shares the launches; and the mixed-integer enumeration built on it must give the reference's regions (tests/test_gpu_mi.py runs
through it by default)."""
import os

import numpy
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

MIXED = ['rand_6_3_12_s1', 'c2_dblint_n5', 'quadtank_n3', 'c4_rand_20_8_20_s0', 'rand_4_2_10_s0', 'dblint_n3', 'rand_5_3_8_s3', 'quadtank_n2',
         'transport_mpqp', 'c2_dblint_n5_x20',
         # open parameter sets (k_recession behind the verdict stages, also inside the shared launches): a pointed cone and one with the
         # main rows' big-M box
         'open_rand_5_3_10_s4_lower', 'open_rand_5_3_10_s4_lower_boxed']


def _levels_alone(g, n_levels, keep_lowdim=False):
    from test_gpu_parity import engine_from_golden
    eng = engine_from_golden(g)
    depth_max = max(eng.n_x, eng.n_t) - eng.n_eq
    depth_max = depth_max if n_levels is None else min(depth_max, n_levels)
    eng.pruned_clear(); eng.frontier_root()
    out = []
    for depth in range(depth_max):
        gen = depth + 1 != depth_max
        st = eng.level_run(gen, keep_lowdim=keep_lowdim)
        out.append(_snapshot(eng, st, gen))
        if not gen or st.n_children == 0:
            break
        eng.frontier_advance()
    eng.close()
    return out


def _snapshot(eng, st, gen):
    """Everything the level left behind.  Region records are keyed by candidate: the slot order and the offsets into the row pool
    come from atomic counters (and the shared launches build the late optimal candidates' regions in the same launch as the others)."""
    hd, hi, er, kk, slots = eng.level_regions_slots()
    regs = {}
    for j in slots.tolist():
        h = numpy.array(hi[j])
        off, n_e = int(h[6]), int(h[2])
        h[6] = 0
        regs[int(h[1])] = (h.tobytes(), numpy.array(hd[j]).tobytes(), numpy.array(er[off:off + n_e]).tobytes())
    return dict(cands=eng.frontier_get().copy(), status=eng.level_status().copy(), regs=regs,
                children=eng.level_children().copy() if gen else None, pruned=eng.level_pruned_new().copy(),
                counts=[int(v) for v in st.n_status], n_children=int(st.n_children))


def _same(a, b, tag):
    assert numpy.array_equal(a['cands'], b['cands']), tag
    assert numpy.array_equal(a['status'], b['status']), tag
    assert a['counts'] == b['counts'] and a['n_children'] == b['n_children'], tag
    assert a['regs'].keys() == b['regs'].keys(), tag
    for c in a['regs']:
        assert a['regs'][c] == b['regs'][c], (tag, c)      # integer head, coefficient head, region rows: bit for bit
    assert sorted(map(tuple, a['pruned'].tolist())) == sorted(map(tuple, b['pruned'].tolist())), tag     # appended through an atomic counter: a set
    if a['children'] is not None:
        assert numpy.array_equal(a['children'], b['children']), tag


@pytest.mark.parametrize('keep_lowdim', [False, True])
def test_batch_levels_equal_the_single_program_levels(keep_lowdim):
    """Ten programs of different shapes (mpQPs with and without equality rows, 4/8-parameter instantiations, one and two tableau
    rows per lane, different depths) advance through shared launches; every member's every level equals its own single-program run.
    Members leave the batch as their frontiers run out."""
    from ppopt_amd import _lib
    from test_gpu_parity import engine_from_golden
    goldens = [load_golden(n) for n in MIXED]
    n_levels = [None if bool(g['complete']) else 3 for g in goldens]
    alone = [_levels_alone(g, nl, keep_lowdim) for g, nl in zip(goldens, n_levels)]
    engs = [engine_from_golden(g) for g in goldens]
    depth_max = [max(e.n_x, e.n_t) - e.n_eq if nl is None else min(max(e.n_x, e.n_t) - e.n_eq, nl) for e, nl in zip(engs, n_levels)]
    for e in engs:
        e.pruned_clear(); e.frontier_root()
    active = list(range(len(engs)))
    depth, shared_total = 0, 0
    while active:
        gens = [depth + 1 != depth_max[i] for i in active]
        stats, n_shared = _lib.Engine.level_run_batch([engs[i] for i in active], gens, keep_lowdim=keep_lowdim)
        shared_total += n_shared
        nxt = []
        for i, st, gen in zip(active, stats, gens):
            _same(_snapshot(engs[i], st, gen), alone[i][depth], (MIXED[i], depth))
            if gen and st.n_children:
                engs[i].frontier_advance()
                nxt.append(i)
        active = nxt
        depth += 1
    assert [len(a) for a in alone] == [min(len(a), depth) for a in alone]
    assert shared_total > 0.5 * sum(len(a) for a in alone)       # the shared launches did the work, not the one-by-one fallback
    for e in engs:
        e.close()


def test_batch_members_that_fall_back_are_run_alone(monkeypatch):
    """MPC_TEST_SMALL_FALLBACK=1: every member reports "repeat on the classic path" after the shared launches (what a level with
    a late optimal candidate or a region the register kernel gives up on does); the call then runs it alone -- same results."""
    from ppopt_amd import _lib
    from test_gpu_parity import engine_from_golden
    names = ['rand_6_3_12_s1', 'quadtank_n3', 'c2_dblint_n5']
    goldens = [load_golden(n) for n in names]
    alone = [_levels_alone(g, 3) for g in goldens]
    monkeypatch.setenv('MPC_TEST_SMALL_FALLBACK', '1')
    engs = [engine_from_golden(g) for g in goldens]
    for e in engs:
        e.pruned_clear(); e.frontier_root()
    stats, n_shared = _lib.Engine.level_run_batch(engs, [True] * 3)
    assert n_shared == 0
    for i, (e, st) in enumerate(zip(engs, stats)):
        _same(_snapshot(e, st, True), alone[i][0], names[i])
        e.close()


def test_batch_memory_budget_runs_the_overflow_alone(monkeypatch):
    """MPC_BATCH_BUDGET_GB: members whose level buffers would not fit the budget next to the others are run after the shared
    launches, one at a time -- same results."""
    from ppopt_amd import _lib
    from test_gpu_parity import engine_from_golden
    names = ['rand_6_3_12_s1', 'quadtank_n3', 'c2_dblint_n5', 'rand_5_3_8_s3']
    goldens = [load_golden(n) for n in names]
    alone = [_levels_alone(g, 2) for g in goldens]
    monkeypatch.setenv('MPC_BATCH_BUDGET_GB', '1e-9')
    engs = [engine_from_golden(g) for g in goldens]
    for e in engs:
        e.pruned_clear(); e.frontier_root()
    stats, n_shared = _lib.Engine.level_run_batch(engs, [True] * len(engs))
    assert n_shared == 1
    for i, (e, st) in enumerate(zip(engs, stats)):
        _same(_snapshot(e, st, True), alone[i][0], names[i])
        e.close()


def _programs():
    import warnings
    from ppopt_amd import MPLP_Program, MPQP_Program, problem_generator as pg
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out = []
        for d in (pg.generate_mpqp_data(6, 3, 12, 1), pg.double_integrator_data(5), pg.quad_tank_data(3), pg.generate_mpqp_data(4, 2, 10, 0),
                  pg.transport_mpqp_data()):
            out.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=d['equality_indices']))
        d = pg.transport_mplp_data()
        out.append(MPLP_Program(d['A'], d['b'], d['c'], d['H'], d['A_t'], d['b_t'], d['F']))
    return out


def test_solve_many_equals_solve():
    """mpqp_hip_combinatorial.solve_many: the Solutions of the shared solve are those of the separate solves (same regions, same
    order, same numbers), base active set included; an mpLP (no register-resident KKT path) shares the call and is run alone."""
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
    one = [mpqp_hip_combinatorial.solve(p) for p in _programs()]
    prof = []
    many = mpqp_hip_combinatorial.solve_many(_programs(), profile=prof)
    assert sum(p['shared_launches'] for p in prof) > 0
    for n, (a, b) in enumerate(zip(one, many)):
        assert len(a.critical_regions) == len(b.critical_regions) > 0, n
        for r1, r2 in zip(a.critical_regions, b.critical_regions):
            assert list(r1.active_set) == list(r2.active_set), n
            assert r1.omega_set == r2.omega_set and r1.lambda_set == r2.lambda_set and r1.regular_set == r2.regular_set, n
            for fld in ('A', 'b', 'C', 'd', 'E', 'f'):
                assert numpy.asarray(getattr(r1, fld)).tobytes() == numpy.asarray(getattr(r2, fld)).tobytes(), (n, fld)


def test_enumeration_batched_equals_one_by_one(monkeypatch):
    """The mixed-integer enumeration with the sub-programs solved together against MPC_NO_BATCH=1 (one handle per fixation, host
    threads): the same regions with the same fixations, bit for bit."""
    import warnings
    from ppopt_amd import MPMIQP_Program
    from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
    from ppopt_amd.problem_generator import generate_mpmiqp_data
    d = generate_mpmiqp_data(6, 3, 12, 4, 1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
    sol_b = solve_mpmiqp(prog)
    monkeypatch.setenv('MPC_NO_BATCH', '1')
    sol_1 = solve_mpmiqp(prog)
    assert len(sol_b.critical_regions) == len(sol_1.critical_regions) > 0
    for r1, r2 in zip(sol_1.critical_regions, sol_b.critical_regions):
        assert list(r1.y_fixation) == list(r2.y_fixation) and list(r1.active_set) == list(r2.active_set)
        for fld in ('A', 'b', 'C', 'd', 'E', 'f'):
            assert numpy.asarray(getattr(r1, fld)).tobytes() == numpy.asarray(getattr(r2, fld)).tobytes(), fld


def test_solve_many_on_random_programs_of_many_shapes():
    """Forty random mpQPs / mpLPs of different sizes in ONE batch -- one to ten parameters (n_theta = 1 has no register-resident
    region kernel: such members run alone inside the call), 3 to 14 variables, with and without equality rows, some infeasible
    from the first level on -- against forty separate solves: the same regions in the same order with the same numbers."""
    import warnings
    from ppopt_amd import MPLP_Program, MPQP_Program, problem_generator as pg
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
    rng = numpy.random.default_rng(77)

    def make():
        out = []
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for j in range(40):
                nx, nt, m = int(rng.integers(3, 15)), int(rng.integers(1, 11)), int(rng.integers(6, 22))
                d = pg.generate_mpqp_data(nx, nt, m, 1000 + j)
                if j % 5 == 4:      # an equality row
                    d['equality_indices'] = [0]
                try:
                    if j % 7 == 6:
                        out.append(MPLP_Program(d['A'], d['b'], d['c'], d['H'], d['A_t'], d['b_t'], d['F'], equality_indices=d['equality_indices']))
                    else:
                        out.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=d['equality_indices']))
                except Exception:
                    continue
        return out
    rng = numpy.random.default_rng(77)
    one = [mpqp_hip_combinatorial.solve(p, max_levels=4) for p in make()]
    rng = numpy.random.default_rng(77)
    progs = make()
    prof = []
    many = mpqp_hip_combinatorial.solve_many(progs, max_levels=4, profile=prof)
    assert len(one) == len(many) >= 30
    assert sum(p['shared_launches'] for p in prof) > sum(p['members'] for p in prof) // 2
    n_regions = 0
    for n, (a, b) in enumerate(zip(one, many)):
        assert len(a.critical_regions) == len(b.critical_regions), n
        key = lambda r: tuple(r.active_set)
        ra, rb = sorted(a.critical_regions, key=key), sorted(b.critical_regions, key=key)      # a level with a late optimal candidate lists it last when solved alone
        for r1, r2 in zip(ra, rb):
            assert list(r1.active_set) == list(r2.active_set), n
            assert r1.omega_set == r2.omega_set and r1.lambda_set == r2.lambda_set and r1.regular_set == r2.regular_set, n
            for fld in ('A', 'b', 'C', 'd', 'E', 'f'):
                assert numpy.asarray(getattr(r1, fld)).tobytes() == numpy.asarray(getattr(r2, fld)).tobytes(), (n, fld)
        n_regions += len(ra)
    assert n_regions > 500


def test_two_batches_at_once_from_two_threads():
    """Two threads, each driving its own batch of programs through shared launches at the same time (two mixed-integer problems
    solved side by side): every call owns its argument table, the results are those of the separate solves."""
    from concurrent.futures import ThreadPoolExecutor
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
    one = [mpqp_hip_combinatorial.solve(p) for p in _programs()]
    sets = [_programs() for _ in range(4)]
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(lambda ps: mpqp_hip_combinatorial.solve_many(ps), sets))
    for many in results:
        for n, (a, b) in enumerate(zip(one, many)):
            assert len(a.critical_regions) == len(b.critical_regions) > 0, n
            for r1, r2 in zip(a.critical_regions, b.critical_regions):
                assert list(r1.active_set) == list(r2.active_set), n
                for fld in ('A', 'b', 'C', 'd', 'E', 'f'):
                    assert numpy.asarray(getattr(r1, fld)).tobytes() == numpy.asarray(getattr(r2, fld)).tobytes(), (n, fld)


def test_solve_many_parks_members_beyond_the_memory_budget(monkeypatch):
    """With a budget that holds one or two members' levels at a time, the others wait at their current level and are resumed when the
    running ones have finished and given their buffers back: same Solutions as the separate solves."""
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
    one = [mpqp_hip_combinatorial.solve(p) for p in _programs()]
    monkeypatch.setenv('MPC_BATCH_BUDGET_GB', '0.002')
    prof = []
    many = mpqp_hip_combinatorial.solve_many(_programs(), profile=prof)
    assert max(p.get('parked', 0) for p in prof) > 0
    for n, (a, b) in enumerate(zip(one, many)):
        assert len(a.critical_regions) == len(b.critical_regions) > 0, n
        for r1, r2 in zip(a.critical_regions, b.critical_regions):
            assert list(r1.active_set) == list(r2.active_set), n
            for fld in ('A', 'b', 'C', 'd', 'E', 'f'):
                assert numpy.asarray(getattr(r1, fld)).tobytes() == numpy.asarray(getattr(r2, fld)).tobytes(), (n, fld)


def test_parked_member_with_large_records_loses_no_region(monkeypatch):
    """a member that produced regions in a level, has children and is then PARKED by the memory admission had its record
    copies only queued -- nothing completed them before its integer heads were read.  Programs with thousands of regions per level
    (copies of several MB, not microseconds) and a budget that parks most members at every level: every Solution must equal the
    separate solve, region for region and byte for byte."""
    import warnings
    from ppopt_amd import MPQP_Program, problem_generator as pg
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial

    def programs():
        out = []
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for seed in (11, 12, 13, 14, 15, 16):
                d = pg.generate_mpqp_data(9, 4, 18, seed)
                out.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F']))
        return out
    one = [mpqp_hip_combinatorial.solve(p) for p in programs()]
    assert max(len(s.critical_regions) for s in one) > 1000
    monkeypatch.setenv('MPC_BATCH_BUDGET_GB', '0.05')
    for _ in range(3):      # (the race needed the copy to lose against the host: a few repetitions)
        prof = []
        many = mpqp_hip_combinatorial.solve_many(programs(), profile=prof)
        assert max(p.get('parked', 0) for p in prof) > 0
        for n, (a, b) in enumerate(zip(one, many)):
            assert len(a.critical_regions) == len(b.critical_regions), n
            ka = sorted((tuple(r.active_set), numpy.asarray(r.A).tobytes(), numpy.asarray(r.E).tobytes()) for r in a.critical_regions)
            kb = sorted((tuple(r.active_set), numpy.asarray(r.A).tobytes(), numpy.asarray(r.E).tobytes()) for r in b.critical_regions)
            assert ka == kb, n

"""Several programs per launch (mpc_level_run_batch): a member's level must be what mpc_level_run computes for
that program alone -- candidates, statuses, region records, children and pruned masks IDENTICAL, bit for bit -- whatever else
shares the launches; and the mixed-integer enumeration built on it must give the reference's regions (tests/test_gpu_mi.py runs
through it by default).

One qualification (round 4).  The region kernel lets several wavefronts share a candidate when the device would otherwise idle; in a
shared launch that number follows the member's SHARE of the launch, not the single program's width, and a different split walks the
facets of a region in a different order.  With the same split on both sides (MPC_NO_RSPLIT=1: one wavefront per candidate) every
record is bit for bit the single program's -- the tests below that say "bit for bit" run that way.  With each side's own split the
x-law, the multipliers, the statuses, the children and the pruned masks are still identical bit for bit; a region's facet list may
differ only where it sits on the LP tolerance -- the CPU oracle's own list for that region changes when its 1e-7 tolerance moves two
decades (conftest.KNIFE_EDGE_FACETS names the one such region of the goldens) -- which the `_own_split` tests check region by region
(tools/batch_w_debug2.py: 14 of 12,871 regions of eight random programs, all of that kind)."""
import os

import numpy
import pytest

from conftest import consume_exception, load_golden

pytestmark = pytest.mark.gpu

MIXED = ['rand_6_3_12_s1', 'c2_dblint_n5', 'quadtank_n3', 'c4_rand_20_8_20_s0', 'rand_4_2_10_s0', 'dblint_n3', 'rand_5_3_8_s3', 'quadtank_n2',
         'transport_mpqp', 'c2_dblint_n5_x20',
         # open parameter sets (k_recession behind the verdict stages, also inside the shared launches): a pointed cone and one with the
         # main rows' big-M box
         'open_rand_5_3_10_s4_lower', 'open_rand_5_3_10_s4_lower_boxed']


@pytest.fixture(autouse=True)
def _same_wavefront_split_on_both_sides(request, monkeypatch):
    """One wavefront per candidate in the region kernel (MPC_NO_RSPLIT=1, read when a handle is created) for every test that
    compares a shared launch with a single-program run bit for bit; the `own_split` tests keep each form's own choice."""
    own = request.node.name.endswith('_own_split') or bool(getattr(getattr(request.node, 'callspec', None), 'params', {}).get('own_split'))
    if not own:
        monkeypatch.setenv('MPC_NO_RSPLIT', '1')


def _facets_on_the_tolerance(prog, active_set):
    """The facet list of this region is a matter of the LP tolerance: the CPU oracle's own list (the reference's one LP per row,
    utils/mpqp_utils.py:143-178) changes when its 1e-7 feasibility tolerance moves two decades either way (the rule of
    conftest.is_knife_edge, applied to the region's index sets)."""
    from oracle import oracle as orc
    P = orc.OracleProblem(prog.A, prog.b, prog.F, prog.c, prog.H, getattr(prog, 'Q', None), prog.A_t, prog.b_t, len(prog.equality_indices))
    lists = set()
    try:
        for tol in (1e-9, 1e-7, 1e-5):
            orc.set_feas_tol(tol)
            v, reg = P.gen_cr_from_active_set(list(active_set))
            lists.add(None if reg is None else (tuple(reg['omega_set']), tuple(reg['lambda_set']), tuple(map(tuple, reg['regular_set']))))
    finally:
        orc.set_feas_tol(1e-7)
    return len(lists) > 1


def _same_regions_up_to_knife_edge_facets(one, many, prog_of, tag):
    """Regions of two Solutions: same active sets; A, b, C, d bit for bit; index sets and E, f bit for bit unless the region's
    facet list sits on the LP tolerance (_facets_on_the_tolerance).  Returns (regions, regions whose facet lists differ)."""
    def key(r):
        fix = getattr(r, 'y_fixation', None)
        return (() if fix is None else tuple(int(v) for v in fix), tuple(r.active_set))
    ra, rb = sorted(one.critical_regions, key=key), sorted(many.critical_regions, key=key)
    assert [key(r) for r in ra] == [key(r) for r in rb], tag
    n_diff = 0
    for r1, r2 in zip(ra, rb):
        for fld in ('A', 'b', 'C', 'd'):
            assert numpy.asarray(getattr(r1, fld)).tobytes() == numpy.asarray(getattr(r2, fld)).tobytes(), (tag, fld, key(r1))
        same = r1.omega_set == r2.omega_set and r1.lambda_set == r2.lambda_set and r1.regular_set == r2.regular_set and all(
            numpy.asarray(getattr(r1, fld)).tobytes() == numpy.asarray(getattr(r2, fld)).tobytes() for fld in ('E', 'f'))
        if not same:
            n_diff += 1
            assert _facets_on_the_tolerance(prog_of(r1), r1.active_set), (tag, key(r1))
    return len(ra), n_diff


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


def _same(a, b, tag, facet_exceptions=None):
    """facet_exceptions = golden name: a region whose facet list differs is accepted if conftest.KNIFE_EDGE_FACETS lists it for
    that golden (x-law and multipliers still bit for bit); None: no exception at all."""
    assert numpy.array_equal(a['cands'], b['cands']), tag
    assert numpy.array_equal(a['status'], b['status']), tag
    assert a['counts'] == b['counts'] and a['n_children'] == b['n_children'], tag
    assert a['regs'].keys() == b['regs'].keys(), tag
    for c in a['regs']:
        if a['regs'][c] != b['regs'][c] and facet_exceptions is not None:
            assert a['regs'][c][1] == b['regs'][c][1], (tag, c)      # coefficient head: x-law and multipliers
            active = a['cands'][c].tolist()
            assert consume_exception('facets', facet_exceptions, active, 'shared launch against single program, different wavefront split'), (tag, c, active)
            continue
        assert a['regs'][c] == b['regs'][c], (tag, c)      # integer head, coefficient head, region rows: bit for bit
    assert sorted(map(tuple, a['pruned'].tolist())) == sorted(map(tuple, b['pruned'].tolist())), tag     # appended through an atomic counter: a set
    if a['children'] is not None:
        assert numpy.array_equal(a['children'], b['children']), tag


@pytest.mark.parametrize('keep_lowdim,own_split', [(False, False), (True, False), (False, True)])
def test_batch_levels_equal_the_single_program_levels(keep_lowdim, own_split):
    """Ten programs of different shapes (mpQPs with and without equality rows, 4/8-parameter instantiations, one and two tableau
    rows per lane, different depths) advance through shared launches; every member's every level equals its own single-program run.
    Members leave the batch as their frontiers run out.  own_split: the region kernel's wavefronts per candidate as each form
    chooses them (the default) instead of one on both sides -- facet lists may then differ on the listed knife-edge regions only."""
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
            _same(_snapshot(engs[i], st, gen), alone[i][depth], (MIXED[i], depth), MIXED[i] if own_split else None)
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


def test_batch_members_that_fall_back_beyond_the_root_level(monkeypatch):
    """a member that ran in the shared launches has had its children written over the previous level's frontier; when it
    then reports "repeat on the classic path", the repeat must not search that buffer for other parents' records (k_xq_thread /
    one-step plans, forced on for every list size here).  Three levels, every one repeated: statuses, regions, children and pruned
    masks are those of the single-program levels."""
    from ppopt_amd import _lib
    from test_gpu_parity import engine_from_golden
    names = ['rand_6_3_12_s1', 'quadtank_n3', 'c2_dblint_n5', 'c4_rand_20_8_20_s0']
    goldens = [load_golden(n) for n in names]
    monkeypatch.setenv('MPC_X1_MIN', '1')
    monkeypatch.setenv('MPC_XQT_MIN', '1')
    alone = [_levels_alone(g, 3) for g in goldens]
    monkeypatch.setenv('MPC_TEST_SMALL_FALLBACK', '1')
    engs = [engine_from_golden(g) for g in goldens]
    for e in engs:
        e.pruned_clear(); e.frontier_root()
    for depth in range(3):
        gens = [depth != 2] * len(engs)
        stats, n_shared = _lib.Engine.level_run_batch(engs, gens)
        assert n_shared == 0
        for i, (e, st) in enumerate(zip(engs, stats)):
            _same(_snapshot(e, st, gens[i]), alone[i][depth], (names[i], depth))
            if gens[i]:
                e.frontier_advance()
    for e in engs:
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


def test_solve_many_with_one_step_plans_in_the_shared_launches(monkeypatch):
    """MPC_BATCH_PLANS=1 (off by default: measured slower on small records): the members' storing levels take the single program's
    one-step plans (k_xq_thread plan mode, k_x1, the rest through k_x2) inside the shared launches.  Same regions, same order, same
    numbers as the separate solves."""
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
    one = [mpqp_hip_combinatorial.solve(p) for p in _programs()]
    monkeypatch.setenv('MPC_BATCH_PLANS', '1')
    monkeypatch.setenv('MPC_BATCH_PLAN_MIN', '1')
    many = mpqp_hip_combinatorial.solve_many(_programs())
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


def test_solve_many_equals_solve_up_to_knife_edge_facets_own_split():
    """The default: the region kernel's wavefronts per candidate follow the member's share of the shared launch (one, when the device
    is full) and the single program's own width (up to eight).  Same regions; x-law and multipliers bit for bit; facet lists bit for
    bit except on regions whose list sits on the LP tolerance, which are few."""
    import warnings
    from ppopt_amd import MPQP_Program, problem_generator as pg
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial

    def programs():
        out = _programs()[:5]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for seed in (21, 22, 23, 24, 25, 26, 27, 28):
                d = pg.generate_mpqp_data(8, 4, 16, seed)
                out.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F']))
        return out
    one = [mpqp_hip_combinatorial.solve(p) for p in programs()]
    progs = programs()
    many = mpqp_hip_combinatorial.solve_many(progs)
    n_regions = n_diff = 0
    for n, (a, b, p) in enumerate(zip(one, many, progs)):
        nr, nd = _same_regions_up_to_knife_edge_facets(a, b, lambda r: p, n)
        n_regions += nr; n_diff += nd
    assert n_regions > 10000 and n_diff <= n_regions // 200, (n_regions, n_diff)


def test_enumeration_batched_equals_one_by_one_up_to_knife_edge_facets_own_split(monkeypatch):
    """The mixed-integer enumeration, sub-programs together against MPC_NO_BATCH=1, each form with its own wavefront split."""
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
        subs = {}

        def sub_of(r):
            fix = tuple(r.y_fixation)
            if fix not in subs:
                subs[fix] = prog.generate_substituted_problem(list(fix))
            return subs[fix]
        nr, nd = _same_regions_up_to_knife_edge_facets(sol_1, sol_b, sub_of, 'mi')
    assert nr > 0 and nd <= max(1, nr // 100), (nr, nd)


def test_wavefront_shares_do_not_change_the_records():
    """The shares (MPC_BATCH_SHARES / MPC_BATCH_WPC_*, read once per process: hence child processes) decide how many wavefronts of a shared
    launch work for a member, not what they compute: with one wavefront per candidate in the region kernel the digest of every region
    of a shared solve is the same without shares, with the default shares and with very narrow ones."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys, warnings\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "warnings.simplefilter('ignore')\n"
        "import numpy\n"
        "from ppopt_amd import MPQP_Program, problem_generator as pg\n"
        "from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m\n"
        "progs = []\n"
        "for seed in (31, 32, 33, 34, 35, 36):\n"
        "    d = pg.generate_mpqp_data(7, 4, 14, seed)\n"
        "    progs.append(MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F']))\n"
        "h = hashlib.sha256(); n = 0\n"
        "for sol in m.solve_many(progs):\n"
        "    for r in sorted(sol.critical_regions, key=lambda r: tuple(r.active_set)):\n"
        "        n += 1\n"
        "        h.update(repr((tuple(r.active_set), tuple(r.omega_set), tuple(r.lambda_set))).encode())\n"
        "        for fld in ('A', 'b', 'C', 'd', 'E', 'f'):\n"
        "            h.update(numpy.asarray(getattr(r, fld)).tobytes())\n"
        "print('DIGEST', n, h.hexdigest())\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for extra in ({'MPC_BATCH_SHARES': '0'}, {}, {'MPC_BATCH_WPC_TH': '1', 'MPC_BATCH_WPC_X2': '1', 'MPC_BATCH_WPC_XQ': '2', 'MPC_BATCH_WPC_R2': '1'}):
        env = dict(os.environ, MPC_NO_RSPLIT='1', **extra)
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith('DIGEST')]
        assert out.returncode == 0 and line, out.stderr[-2000:]
        digests.append(line[0])
    assert int(digests[0].split()[1]) > 1000
    assert digests[0] == digests[1] == digests[2], digests

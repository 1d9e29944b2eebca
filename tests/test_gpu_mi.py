"""Mixed-integer caller of the path on the MI355X against goldens captured from the
reference's own mixed-integer test problems (tests/golden/mi_*.npz, oracle/ref_harness/gen_mi_goldens.py).

Checked per problem: the mixed-integer presolve, every partial binary fixation, the list of feasible fixations, every
substituted continuous program and its region set, the final Solution (after the 1-D overlap reduction where the
reference applies it) and Solution.evaluate / evaluate_objective at sampled parameter points.  The region counts the
reference's tests assert (tests/mpmiqp_solver_tests/test_mpmiqp.py:86-140) are restated at the end."""
import glob
import os
import warnings

import numpy
import pytest

from conftest import rows_match

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MI_FILES = sorted(glob.glob(os.path.join(GOLDEN, 'mi_*.npz')))
IDS = [os.path.basename(p)[3:-4] for p in MI_FILES]
TOL = 1e-8


def build(g, post_process=True):
    from ppopt_amd import MPMILP_Program, MPMIQP_Program
    kw = {}
    for key in ('c_c', 'c_t', 'Q_t'):
        if 'raw_' + key in g.files:
            kw[key] = g['raw_' + key]
    if 'raw_equality_indices' in g.files:
        kw['equality_indices'] = g['raw_equality_indices'].tolist()
    bins = g['raw_binary_indices'].tolist()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if str(g['cls']) == 'MPMIQP_Program':
            return MPMIQP_Program(g['raw_A'], g['raw_b'], g['raw_c'], g['raw_H'], g['raw_Q'], g['raw_A_t'],
                                  g['raw_b_t'], g['raw_F'], bins, post_process=post_process, **kw)
        return MPMILP_Program(g['raw_A'], g['raw_b'], g['raw_c'], g['raw_H'], g['raw_A_t'], g['raw_b_t'], g['raw_F'],
                              bins, post_process=post_process, **kw)


def region_map(g, prefix):
    out = {}
    for i in range(int(g[prefix + 'n'])):
        k, ne = int(g[prefix + 'k'][i]), int(g[prefix + 'nE'][i])
        out[tuple(g[prefix + 'as'][i, :k].tolist())] = dict(
            A=g[prefix + 'A'][i], b=g[prefix + 'b'][i], C=g[prefix + 'C'][i, :k], d=g[prefix + 'd'][i, :k],
            E=g[prefix + 'E'][i, :ne], f=g[prefix + 'f'][i, :ne])
    return out


def same_rows(E1, f1, E2, f2):
    """Row sets equal under 1e-8, order and numerically duplicate rows ignored: the reference removes only EXACT
    duplicates (constraint_utilities.py:125-134), so whether two copies of a facet that differ in the last bit
    collapse is rounding noise of either side, not a property of the region."""
    return rows_match(E1, f1, E2, f2, TOL)


@pytest.mark.parametrize('path', MI_FILES, ids=IDS)
def test_mixed_integer_presolve(path):
    g = numpy.load(path)
    prog = build(g)
    for key in ('A', 'b', 'F', 'A_t', 'b_t', 'c', 'H'):
        assert getattr(prog, key).shape == g['proc_' + key].shape, key
        numpy.testing.assert_allclose(getattr(prog, key), g['proc_' + key], atol=1e-12, rtol=0, err_msg=key)
    assert list(prog.equality_indices) == g['proc_eq'].tolist()
    assert prog.binary_indices == g['binary_indices'].tolist()
    assert prog.cont_indices == g['cont_indices'].tolist()


@pytest.mark.parametrize('path', MI_FILES, ids=IDS)
def test_binary_fixations(path):
    from ppopt_amd.mp_solvers.mitree import MITree
    g = numpy.load(path)
    prog = build(g)
    for fix, want in zip(g['bin_fix'].tolist(), g['bin_feas'].tolist()):
        partial = [v for v in fix if v >= 0]
        assert prog.check_bin_feasibility(partial) == want, partial
    assert prog.feasible_combinations() == g['combos'].tolist()
    tree = MITree(prog, depth=0)
    assert tree.count_nodes() == int(g['n_nodes'])
    assert [leaf.fixed_bins for leaf in tree.get_full_leafs()] == g['combos'].tolist()


@pytest.mark.parametrize('path', MI_FILES, ids=IDS)
def test_substituted_programs_and_their_regions(path):
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    g = numpy.load(path)
    prog = build(g)
    for i, fix in enumerate(g['combos'].tolist()):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            sub = prog.generate_substituted_problem(fix)
        for key in ('A', 'b', 'F', 'A_t', 'b_t', 'c', 'H', 'c_c', 'c_t'):
            want = g[f'S{i}_{key}']
            assert getattr(sub, key).shape == want.shape, (fix, key)
            numpy.testing.assert_allclose(getattr(sub, key), want, atol=1e-12, rtol=0, err_msg=f'{fix} {key}')
        if f'S{i}_Q' in g.files:
            numpy.testing.assert_allclose(sub.Q, g[f'S{i}_Q'], atol=1e-12, rtol=0)
        assert list(sub.equality_indices) == g[f'S{i}_eq'].tolist()
        sol = solve_mpqp(sub, mpqp_algorithm.combinatorial)
        want = region_map(g, f'S{i}_R_')
        got = {tuple(r.active_set): r for r in sol.critical_regions}
        assert sorted(got) == sorted(want), fix
        for key, ref in want.items():
            r = got[key]
            numpy.testing.assert_allclose(r.A, ref['A'], atol=TOL, rtol=0)
            numpy.testing.assert_allclose(r.b.flatten(), ref['b'], atol=TOL, rtol=0)
            numpy.testing.assert_allclose(r.C, ref['C'], atol=TOL, rtol=0)
            numpy.testing.assert_allclose(r.d.flatten(), ref['d'], atol=TOL, rtol=0)
            assert same_rows(r.E, r.f, ref['E'], ref['f']), (fix, key)


@pytest.mark.parametrize('path', MI_FILES, ids=IDS)
def test_solve_mpmiqp_matches_reference(path):
    from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
    g = numpy.load(path)
    prog = build(g)
    sol = solve_mpmiqp(prog, num_cores=1)
    assert sol.is_mixed_integer_sol()
    assert len(sol) == int(g['F_n'])
    assert bool(sol.is_overlapping) == bool(g['F_overlapping'])
    want = []
    for i in range(int(g['F_n'])):
        k, ne = int(g['F_k'][i]), int(g['F_nE'][i])
        want.append((tuple(g['F_y'][i].tolist()), tuple(g['F_as'][i, :k].tolist()), g['F_E'][i, :ne], g['F_f'][i, :ne],
                     g['F_A'][i], g['F_b'][i]))
    used = set()
    for r in sol.critical_regions:
        assert r.y_indices == prog.binary_indices and r.x_indices == prog.cont_indices
        hit = None
        for j, (y, aset, E, f, A, b) in enumerate(want):
            if j in used or y != tuple(r.y_fixation) or aset != tuple(r.active_set):
                continue
            if same_rows(r.E, r.f, E, f) and numpy.allclose(r.A, A, atol=TOL) and numpy.allclose(r.b.flatten(), b, atol=TOL):
                hit = j
                break
        assert hit is not None, (r.y_fixation, r.active_set, r.E, r.f)
        used.add(hit)
    assert len(used) == len(want)
    # the evaluation the reference's tests exercise: Solution.evaluate / evaluate_objective
    for th, ok, x, obj in zip(g['T_theta'], g['T_ok'], g['T_x'], g['T_obj']):
        got = sol.evaluate(th.reshape(-1, 1))
        assert (got is not None) == bool(ok), th
        if ok:
            assert abs(sol.evaluate_objective(th.reshape(-1, 1)) - obj) <= 1e-7 * max(1.0, abs(obj)), th
            if not bool(g['F_overlapping']):
                numpy.testing.assert_allclose(got.flatten(), x, atol=1e-7, rtol=1e-9)


def _load(name):
    return numpy.load(os.path.join(GOLDEN, f'mi_{name}.npz'))


def test_region_counts_the_reference_asserts():
    """tests/mpmiqp_solver_tests/test_mpmiqp.py:86-132."""
    from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
    from ppopt_amd.utils.region_overlap_utils import get_bounds_1d
    for name, count in (('bard_mpMILP_adapted_degenerate', 5), ('mpMILP_1d', 3), ('pappas_multi_objective', 3)):
        assert len(solve_mpmiqp(build(_load(name)), num_cores=1)) == count, name
    sol = solve_mpmiqp(build(_load('bard_mpMILP_adapted_degenerate')), num_cores=1)
    assert numpy.isclose(sol.evaluate_objective(numpy.array([[2.]])), 2)
    assert numpy.isclose(sol.evaluate_objective(numpy.array([[3.]])), 1)
    sol = solve_mpmiqp(build(_load('bard_mpMILP_adapted_2')), num_cores=1)
    for th, val in ((2., 2), (3., 1), (8., 1), (9., 3)):
        assert numpy.isclose(sol.evaluate_objective(numpy.array([[th]])), val)
    sol = solve_mpmiqp(build(_load('mpMILP_1d')), num_cores=1)
    for th, val in ((2., 2), (45., 40), (60., 50)):
        assert numpy.isclose(sol.evaluate_objective(numpy.array([[th]])), val)
    for cr in sol.critical_regions:
        lb, ub = get_bounds_1d(cr.E, cr.f)
        assert any(numpy.isclose(lb, a) and numpy.isclose(ub, b) for a, b in [(0, 40), (40, 50), (50, 100)])


def test_explicit_solution_agrees_with_the_milp_at_a_point():
    """tests/mpmiqp_solver_tests/test_mpmiqp.py:60-76,134-155: evaluate() against the MILP solved at that theta
    (on the device, as the batch of LPs over the fixations)."""
    from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
    for name, theta in (('mpMILP_market_problem', [[0.0], [500.0]]), ('acevedo_mpmilp', [[0.5]] * 3),
                        ('pappas_multi_objective_2', [[90.0]])):
        prog = build(_load(name))
        sol = solve_mpmiqp(prog, num_cores=1)
        th = numpy.array(theta)
        det = prog.solve_theta(th)
        assert det is not None
        assert numpy.isclose(det.obj, sol.evaluate_objective(th)), name
        assert numpy.allclose(det.sol, sol.evaluate(th).flatten(), atol=1e-6), name


def test_sub_problem_shape_and_wrong_algorithm():
    """test_mpmiqp.py:20-25,78-84."""
    from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
    prog = build(_load('simple_mpMILP'))
    sub = prog.generate_substituted_problem([0, 1])
    assert sub.A.shape == (2, 1) and sub.equality_indices == [0]
    assert prog.check_bin_feasibility([0, 0]) and prog.check_bin_feasibility([1, 0])
    assert prog.check_bin_feasibility([0, 1]) and not prog.check_bin_feasibility([1, 1])
    with pytest.raises(TypeError):
        solve_mpmiqp(prog, 'enum')
    solve_mpmiqp(build(_load('simple_mpMIQP')), num_cores=1).evaluate(numpy.array([[1.2]]))


@pytest.mark.parametrize('path', MI_FILES, ids=IDS)
def test_batched_point_location_on_mixed_integer_solutions(path):
    """Solution.evaluate_batch on a mixed-integer solution (full variable vector, binaries spliced in; overlapping
    regions resolved by the objective on the device) against the host loop of get_region / evaluate."""
    from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
    g = numpy.load(path)
    prog = build(g)
    sol = solve_mpmiqp(prog, num_cores=1)
    thetas = g['T_theta']
    x_b, idx_b = sol.evaluate_batch(thetas)
    assert x_b.shape == (len(thetas), prog.num_x())
    for th, xb, ib in zip(thetas, x_b, idx_b):
        cr = sol.get_region(th.reshape(-1, 1))
        if cr is None:
            assert ib == -1 and numpy.all(numpy.isnan(xb))
            continue
        want = cr.evaluate(th.reshape(-1, 1)).flatten()
        got_obj = prog.evaluate_objective(xb.reshape(-1, 1), th.reshape(-1, 1))
        want_obj = prog.evaluate_objective(want.reshape(-1, 1), th.reshape(-1, 1))
        assert ib >= 0
        assert abs(got_obj - want_obj) <= 1e-7 * max(1.0, abs(want_obj))       # same objective; ties may pick a twin
        if sol.critical_regions[int(ib)] is cr:
            numpy.testing.assert_allclose(xb, want, atol=1e-8, rtol=1e-10)


def test_upop_point_location_object():
    """upop.PointLocation (tests/upop_tests/test_point_location.py:9-32): evaluate() equals Solution.evaluate(), with
    and without the overlap flag; exact E theta <= f semantics; batch and single-point forms agree."""
    import copy
    from ppopt_amd import MPQP_Program
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    from ppopt_amd.problem_generator import transport_mpqp_data
    from ppopt_amd.upop import PointLocation
    d = transport_mpqp_data()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
    sol = solve_mpqp(prog, mpqp_algorithm.combinatorial)
    theta = numpy.array([[200.0], [200.0]])
    for overlapping in (False, True):
        s2 = copy.copy(sol)
        s2.is_overlapping = overlapping
        pl = PointLocation(s2)
        assert pl.is_inside(theta)
        assert numpy.allclose(pl.evaluate(theta), sol.evaluate(theta))
        assert sol.critical_regions[pl.locate(theta)].is_inside(theta)
    pl = PointLocation(sol)
    rng = numpy.random.default_rng(3)
    pts = rng.uniform(-50, 1100, size=(500, 2))
    idx = pl.locate_batch(pts)
    x, idx2 = pl.evaluate_batch(pts)
    assert numpy.array_equal(idx, idx2)
    for p, i in zip(pts[:60], idx[:60]):
        inside = [j for j, r in enumerate(sol.critical_regions) if numpy.all(r.E @ p.reshape(-1, 1) <= r.f)]
        assert (i == -1 and not inside) or (i == inside[0])
    assert pl.evaluate(numpy.array([[-500.0], [-500.0]])) is None and not pl.is_inside(numpy.array([[-500.0], [-500.0]]))


def test_upop_point_location_boundary_points_are_inside():
    """`A @ theta <= b` of the reference's PointLocation (upop/point_location.py:46,59) is inclusive: points exactly on a
    facet, on a vertex of the parameter box and theta = 0 on a `-theta <= 0` row are located (the locator's strict
    `E theta - f < tol` with tol = 0 would report -1 for all of them), and Solution.get_region keeps the strict test."""
    from ppopt_amd import MPQP_Program
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    from ppopt_amd.problem_generator import transport_mpqp_data
    from ppopt_amd.upop import PointLocation
    d = transport_mpqp_data()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
    sol = solve_mpqp(prog, mpqp_algorithm.combinatorial)
    pl = PointLocation(sol)

    def host(theta):   # the reference's get_region_no_overlap, on the host
        for j, r in enumerate(sol.critical_regions):
            if numpy.all(r.E @ theta.reshape(-1, 1) <= r.f):
                return j
        return -1

    # the vertices of the parameter box {A_t theta <= b_t} that the program leaves feasible, and theta = 0
    pts = [numpy.zeros(2)]
    At, bt = prog.A_t, prog.b_t.ravel()
    for i in range(len(bt)):
        for j in range(i + 1, len(bt)):
            M = At[[i, j]]
            if abs(numpy.linalg.det(M)) > 1e-9:
                v = numpy.linalg.solve(M, bt[[i, j]])
                if numpy.all(At @ v <= bt + 1e-9):
                    pts.append(v)
    # points exactly on a facet: a vertex of a region (two of its rows tight), axis-aligned rows give exact arithmetic
    for r in sol.critical_regions:
        E, f = numpy.asarray(r.E), numpy.asarray(r.f).ravel()
        for i in range(len(f)):
            nz = numpy.flatnonzero(numpy.abs(E[i]) > 1e-12)
            if len(nz) == 1:                   # axis-aligned facet  e * theta_t <= f
                t = nz[0]
                base = numpy.array([200.0, 200.0])
                base[t] = f[i] / E[i, t]
                pts.append(base)
    pts = numpy.array(pts)
    got = pl.locate_batch(pts)
    want = numpy.array([host(p) for p in pts])
    assert numpy.array_equal(got, want), (got, want)
    assert (want >= 0).sum() >= 3                                     # the boundary points really are inside something
    strict = pl._exact.get_region_batch(pts)                          # same tolerance 0, strict test
    assert (strict[want >= 0] == -1).any()                            # ... and the strict test would have lost some
    # single-point API on the box vertex theta = 0 (row -theta <= 0 tight)
    assert pl.is_inside(numpy.zeros((2, 1))) == (host(numpy.zeros(2)) >= 0)
    assert pl.locate(numpy.zeros((2, 1))) == host(numpy.zeros(2))
    x0 = pl.evaluate(numpy.zeros((2, 1)))
    if host(numpy.zeros(2)) >= 0:
        assert numpy.allclose(x0, sol.critical_regions[host(numpy.zeros(2))].evaluate(numpy.zeros((2, 1))))
    # Solution.get_region stays strict with its own tolerance (solution.py:60-112, critical_region.py:83-86)
    th = pts[:8]
    idx = sol.get_region_batch(th)
    for p, i in zip(th, idx):
        cr = sol.get_region(p.reshape(-1, 1))
        assert (cr is None and i == -1) or sol.critical_regions[int(i)] is cr


def test_program_generators_return_solvable_programs():
    """generate_mpqp / generate_mplp as in the reference's problem_generator.py (tests/other_tests/test_problem_generator.py)."""
    from ppopt_amd import MPLP_Program, MPQP_Program
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    from ppopt_amd.problem_generator import generate_mplp, generate_mpqp
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        qp = generate_mpqp(4, 2, 10, seed=0)
        lp = generate_mplp(4, 2, 10, seed=0)
    assert isinstance(qp, MPQP_Program) and isinstance(lp, MPLP_Program) and not isinstance(lp, MPQP_Program)
    assert qp.num_x() == 4 and qp.num_t() == 2 and lp.num_x() == 4
    g = numpy.load(os.path.join(GOLDEN, 'rand_4_2_10_s0.npz'))
    numpy.testing.assert_allclose(qp.A, g['proc_A'], atol=1e-12)
    assert len(solve_mpqp(qp, mpqp_algorithm.combinatorial)) == int(len(g['R_k']))
    assert len(solve_mpqp(lp, mpqp_algorithm.combinatorial)) > 0
    # the reference's own checks (tests/other_tests/test_problem_generator.py:4-11)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert generate_mplp(2, 2, 40, seed=3).feasible_theta_point() is not None
        assert generate_mpqp(2, 2, 40, seed=3).feasible_theta_point() is not None


def test_sampling_helpers_of_the_program_classes():
    """solve_theta_variable / solve_theta_batch / gen_optimal_active_set / sample_theta_space (mplp_program.py:355-664)
    on the transport mpLP of the tutorial: every sampled active set belongs to a region of the solved problem."""
    from ppopt_amd import MPLP_Program
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    from ppopt_amd.problem_generator import transport_mplp_data
    d = transport_mplp_data()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPLP_Program(d['A'], d['b'], d['c'], d['H'], d['A_t'], d['b_t'], d['F'])
    sol = solve_mpqp(prog, mpqp_algorithm.combinatorial)
    region_sets = {tuple(r.active_set) for r in sol.critical_regions}
    assert prog.solve_theta_variable() is not None
    th = prog.feasible_theta_point()
    one = prog.solve_theta(th)
    many = prog.solve_theta_batch(numpy.vstack([th.T, th.T + 1e9]))
    assert many[1] is None and abs(many[0].obj - one.obj) <= 1e-9 * max(1.0, abs(one.obj))
    numpy.testing.assert_allclose(many[0].sol, one.sol, atol=1e-9)
    numpy.testing.assert_allclose(sol.evaluate(th).flatten(), one.sol, atol=1e-7)
    a = prog.gen_optimal_active_set()
    assert a is not None and tuple(a) in region_sets
    walked = prog.sample_theta_space(20)
    assert walked and all(len(w) >= prog.num_x() for w in walked)


def test_verify_solution_and_verify_theta_through_the_kkt_conditions():
    """Solution.verify_solution / verify_theta (solution.py:114-174) without a QP solver: KKT conditions at the Chebyshev
    centres (one LP batch) and at sampled points; a corrupted law must be caught."""
    from ppopt_amd import MPQP_Program
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    from ppopt_amd.problem_generator import generate_mpqp_data, transport_mpqp_data
    for d in (transport_mpqp_data(), generate_mpqp_data(6, 3, 12, 1)):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
        sol = solve_mpqp(prog, mpqp_algorithm.combinatorial)
        centres, radii = sol.chebyshev_centres()
        assert numpy.all(radii > 1e-8)
        for cr, c in zip(sol.critical_regions[:10], centres[:10]):
            assert cr.is_inside(c.reshape(-1, 1), 0.0)
        assert sol.verify_solution()
        assert all(sol.verify_theta(c.reshape(-1, 1)) for c in centres[:20])
        assert sol.verify_theta(numpy.full((prog.num_t(), 1), 1e9))          # outside Theta
        bad = sol.critical_regions[0]
        bad.b = numpy.asarray(bad.b) + 1e-3
        assert not sol.verify_solution()


@pytest.mark.parametrize('shape', [(5, 3, 10, 4), (6, 2, 14, 7), (4, 4, 9, 11)], ids=['5_3_10', '6_2_14', '4_4_9'])
def test_regions_cover_exactly_the_feasible_parameters(shape):
    """Completeness property of a fully solved program, no oracle involved: a parameter point lies in some critical
    region iff the program is feasible there (one LP per point, all points one device batch); points within 1e-4 of the
    boundary of the feasible set are not judged."""
    from ppopt_amd import MPQP_Program, _lib
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    from ppopt_amd.problem_generator import generate_mpqp_data
    nx, nt, m, seed = shape
    d = generate_mpqp_data(nx, nt, m, seed)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
    sol = solve_mpqp(prog, mpqp_algorithm.combinatorial)
    assert len(sol) > 0 and sol.verify_solution()
    rng = numpy.random.default_rng(seed)
    box = float(numpy.max(numpy.abs(prog.b_t)))
    pts = rng.uniform(-box, box, size=(4000, nt))
    pts = pts[numpy.all(pts @ prog.A_t.T <= prog.b_t.reshape(1, -1), axis=1)]
    located = sol.get_region_batch(pts) >= 0

    def feasible(shrink):
        rhs = prog.b.reshape(1, -1) + pts @ prog.F.T - shrink
        flags = numpy.zeros((len(pts), prog.num_constraints()), dtype=numpy.uint8)
        status, _, _, _ = _lib.lp_solve_batch(prog.A, rhs, None, flags)
        return status == _lib.LP_OPTIMAL

    inner, outer = feasible(1e-4), feasible(-1e-4)
    assert inner.sum() > 100
    assert numpy.all(located[inner]), 'a feasible parameter point lies in no region'
    assert not numpy.any(located[~outer]), 'a region contains an infeasible parameter point'


def test_failed_create_does_not_leak_device_memory():
    """mpc_create that fails after its first allocations (a program whose tableau exceeds the 160 KiB LDS) gives every
    block, stream and event back: repeating it does not consume device memory."""
    import torch
    from ppopt_amd import _lib
    rng = numpy.random.default_rng(0)
    nx, nt, nc, ntc = 200, 8, 128, 16
    A, b, F = rng.standard_normal((nc, nx)), numpy.ones(nc), rng.standard_normal((nc, nt))
    At = numpy.vstack([numpy.eye(nt), -numpy.eye(nt)])
    args = (A, b, F, numpy.zeros(nx), numpy.zeros((nx, nt)), numpy.eye(nx), At, numpy.ones(ntc), 0)

    def attempt():
        with pytest.raises(_lib.MpcError, match='LDS'):
            _lib.Engine(*args)

    for _ in range(3):
        attempt()                                  # fills the recycling pools
    free0, _ = torch.cuda.mem_get_info(0)
    for _ in range(40):
        attempt()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 <= (8 << 20), (free0, free1)


def test_solver_device_is_the_device_of_the_solve():
    """Solver(device=d): presolve LPs, the engine and solve_mpqp all use device d (here d = 0 explicitly, and an
    out-of-range device fails loudly instead of silently landing on GPU 0)."""
    from ppopt_amd import MPQP_Program, Solver, _lib
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    from ppopt_amd.problem_generator import transport_mpqp_data
    d = transport_mpqp_data()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], solver=Solver(device=0))
    sol = solve_mpqp(prog, mpqp_algorithm.combinatorial)
    assert len(sol.critical_regions) == 4 and prog.engine().device == 0
    n_dev = _lib.load().mpc_device_count()
    prog.solver = Solver(device=n_dev)             # one past the last device
    prog.release_engine()
    with pytest.raises(_lib.MpcError):
        solve_mpqp(prog, mpqp_algorithm.combinatorial)


def test_many_binaries_stay_within_the_batch_budget():
    """14 binaries under a cardinality constraint: the leaf table is built level by level with relaxation pruning (470
    feasible leaves of 16,384), presolve and solve_theta run in batches bounded by MILP_BATCH_BYTES."""
    from ppopt_amd import Solver
    nb = 14
    n = 1 + nb
    A = numpy.zeros((3, n))
    A[0, 1:] = 1.0
    A[1, 0] = 1.0
    A[1, 1:] = -(0.5 ** numpy.arange(nb))
    A[2, 0] = -1.0
    b = numpy.array([3.0, 0.0, 0.0])
    S = Solver()
    S.MILP_BATCH_BYTES = 64 << 10
    table = S.milp_leaf_feasibility(A, b, [], list(range(1, n)))
    ones = numpy.array([bin(i).count('1') for i in range(1 << nb)])
    assert numpy.array_equal(table, ones <= 3) and int(table.sum()) == 470
    got = S.milp_any_feasible(A, b, [[0], [1], [2], [0, 2]], list(range(1, n)), numpy.flatnonzero(table))
    assert got.tolist() == [True, True, True, True]
    c = numpy.zeros(n)
    c[0] = -1.0
    out = S.solve_milp(c, A, b, [], list(range(1, n)))
    assert abs(out.obj + 1.75) <= 1e-9 and out.sol[1:].tolist() == [1, 1, 1] + [0] * (nb - 3)

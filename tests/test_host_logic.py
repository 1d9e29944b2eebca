"""Host layer on CPU: presolve parity with the reference (golden vectors), constraint utilities, generators, result
objects, dispatch.  Programs are built with the oracle as the deterministic-solver plug (``solver=`` argument of the
constructor, the reference's own plug point) because the product LP backend needs the GPU."""
import numpy
import pytest

from conftest import load_golden
from ppopt_amd import MPLP_Program, MPQP_Program, CriticalRegion, Solution
from ppopt_amd import problem_generator as pg
from ppopt_amd.utils import constraint_utilities as cu
from ppopt_amd.utils.general_utils import make_column, ppopt_block, select_not_in_list

PRESOLVE = ['c1_transport_mplp', 'mplp_rand_4_2_10_s0', 'mplp_rand_5_3_12_s2', 'transport_mpqp', 'dblint_n3', 'c2_dblint_n5', 'rand_4_2_10_s0', 'rand_5_3_8_s3',
            'rand_6_3_12_s1', 'quadtank_n2', 'quadtank_n3', 'c5_control_allocation', 'c4_rand_20_8_20_s0',
            'c3_quadtank_n10']


def build_program(g, solver):
    raw = {k[4:]: g[k] for k in g.files if k.startswith('raw_')}
    eq = [int(v) for v in raw['eq']]
    if bool(g['is_mplp']):
        return MPLP_Program(raw['A'], raw['b'], raw['c'], raw['H'], raw['A_t'], raw['b_t'], raw['F'],
                            equality_indices=eq, solver=solver)
    return MPQP_Program(raw['A'], raw['b'], raw['c'], raw['H'], raw['Q'], raw['A_t'], raw['b_t'], raw['F'],
                        equality_indices=eq, solver=solver)


@pytest.mark.parametrize('name', PRESOLVE)
def test_presolve_matches_reference(oracle, name):
    """raw constructor inputs -> processed (A, b, F, A_t, b_t, equality_indices) exactly as the reference's
    constructor produces them (mplp_program.py:60-134, 285-306): this fixes the meaning of every active-set index."""
    g = load_golden(name)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = build_program(g, oracle.OracleSolver())
    assert prog.equality_indices == [int(v) for v in g['proc_eq']]
    for fld in ('A', 'b', 'F', 'A_t', 'b_t'):
        ref = g['proc_' + fld]
        got = getattr(prog, fld)
        assert got.shape == ref.shape, fld
        assert numpy.allclose(got, ref, rtol=1e-13, atol=1e-13), fld


def test_generators_reproduce_golden_inputs():
    """The problem builders regenerate, bit for bit, the raw matrices the goldens were produced from."""
    cases = {'rand_4_2_10_s0': pg.generate_mpqp_data(4, 2, 10, 0), 'rand_6_3_12_s1': pg.generate_mpqp_data(6, 3, 12, 1),
             'c4_rand_20_8_20_s0': pg.generate_mpqp_data(20, 8, 20, 0), 'c2_dblint_n5': pg.double_integrator_data(5),
             'c3_quadtank_n10': pg.quad_tank_data(10), 'c5_control_allocation': pg.control_allocation_data(),
             'c1_transport_mplp': pg.transport_mplp_data(), 'transport_mpqp': pg.transport_mpqp_data()}
    for name, d in cases.items():
        g = load_golden(name)
        for fld in ('A', 'b', 'c', 'H', 'Q', 'A_t', 'b_t', 'F'):
            if d[fld] is None:
                assert 'raw_' + fld not in g.files
            else:
                assert numpy.array_equal(d[fld], g['raw_' + fld]), (name, fld)


def test_scale_and_zero_rows():
    """other_tests/test_constraint_utilities.py: scaling to unit norm, zero / duplicate row handling"""
    A = numpy.array([[3.0, 4.0], [0.0, 2.0]]); b = numpy.array([[5.0], [4.0]])
    As, bs = cu.scale_constraint(A, b)
    assert numpy.allclose(numpy.linalg.norm(As, axis=1), 1) and numpy.allclose(bs.ravel(), [1.0, 2.0])
    A = numpy.array([[1.0, 0], [0, 0], [1.0, 0], [0, 1e-9]]); b = numpy.array([[1.0], [2.0], [1.0], [3.0]])
    A1, b1 = cu.remove_zero_rows(A, b)
    assert A1.shape[0] == 3
    A2, b2 = cu.remove_duplicate_rows(A, b)
    assert A2.shape[0] == 3 and numpy.array_equal(A2[0], [1.0, 0])
    assert cu.numerically_nonzero_rows(A) == [0, 2]
    A3, b3 = cu.remove_numerically_zero_rows(A, b)
    assert A3.shape[0] == 2
    kept, gone = cu.get_indices_of_zero_rows(A)
    assert kept == [0, 2] and gone == [1, 3]


def test_implicit_equalities_and_reduction():
    """other_tests/test_constraint_utilities.py:153-196 style known answers"""
    A = numpy.array([[1.0, 1.0], [-1.0, -1.0], [1.0, 0.0], [2.0, 0.0]]); b = numpy.array([[1.0], [-1.0], [3.0], [6.0]])
    assert cu.detect_implicit_equalities(A, b) == [[0, 1]]
    F = numpy.zeros((4, 1))
    A2, b2, F2, eq = cu.find_implicit_equalities(A, b, F, [])
    assert eq == [0] and A2.shape[0] == 3 and numpy.array_equal(A2[0], [1.0, 1.0])
    # dependent equalities are reduced (the reference's loop never examines the last row)
    Ae = numpy.array([[1.0, 0.0], [2.0, 0.0], [0.0, 1.0], [1.0, 1.0]]); be = numpy.ones((4, 1)); Fe = numpy.zeros((4, 1))
    A3, b3, F3, eq3 = cu.generate_reduced_equality_constraints(Ae, be, Fe, [0, 1, 2])
    assert eq3 == [0] and A3.shape[0] == 2
    # rows without x move to the parametric block
    A4 = numpy.array([[1.0, 0.0], [0.0, 0.0]]); b4 = numpy.array([[1.0], [2.0]]); F4 = numpy.array([[0.0], [1.0]])
    A5, b5, F5, At5, bt5 = cu.process_program_constraints(A4, b4, F4, numpy.array([[1.0]]), numpy.array([[5.0]]))
    assert A5.shape[0] == 1 and At5.shape[0] == 2 and numpy.allclose(At5[1], [-1.0]) and bt5[1, 0] == 2.0


def test_block_helpers():
    a, b = numpy.ones((2, 2)), numpy.zeros((2, 1))
    assert numpy.array_equal(ppopt_block([[a, b], [b.T @ a, numpy.array([[7.0]])]]), numpy.block([[a, b], [b.T @ a, numpy.array([[7.0]])]]))
    assert numpy.array_equal(ppopt_block([a, b]), numpy.hstack([a, b]))
    assert make_column([1, 2, 3]).shape == (3, 1)
    assert select_not_in_list(numpy.arange(10).reshape(5, 2), [1, 3]).tolist() == [[0, 1], [4, 5], [8, 9]]


def test_critical_region_and_solution():
    """other_tests/test_critical_region.py, mpqp_solver_tests/test_solution.py: the square region of test_fixtures.py:66-75"""
    E = numpy.vstack([numpy.eye(2), -numpy.eye(2)]); f = make_column([1, 1, 0, 0])
    cr = CriticalRegion(numpy.eye(2), numpy.zeros((2, 1)), numpy.eye(2), numpy.zeros((2, 1)), E, f, [])
    th = make_column([0.5, 0.5])
    assert numpy.array_equal(cr.evaluate(th), th) and numpy.array_equal(cr.lagrange_multipliers(th), th)
    assert cr.is_inside(th) and not cr.is_inside(make_column([2.0, 0.5]))
    assert cr.get_constraints()[0] is E and 'active set []' in repr(cr)

    class _P:
        def num_t(self): return 2
        def evaluate_objective(self, x, t): return float((x.T @ x)[0, 0])
    sol = Solution(_P(), [])
    assert len(sol) == 0 and sol.evaluate(th) is None and sol.get_region(th) is None
    sol.add_region(cr)
    assert len(sol) == 1 and sol.get_region(th) is cr and numpy.array_equal(sol.evaluate(th), th)
    sol.is_overlapping = True
    cr2 = CriticalRegion(0.5 * numpy.eye(2), numpy.zeros((2, 1)), numpy.eye(2), numpy.zeros((2, 1)), E, f, [1])
    sol.add_region(cr2)
    assert sol.get_region(th) is cr2 and sol.evaluate_objective(th) == pytest.approx(0.125)
    assert sol.theta_dim() == 2


def test_solve_mpqp_rejects_non_enum_and_out_of_scope():
    """other_tests/test_solve_mpqp.py:94-100: a string instead of the enum raises TypeError"""
    from ppopt_amd.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp
    with pytest.raises(TypeError):
        solve_mpqp(None, algorithm='cambinatorial')
    with pytest.raises(NotImplementedError):
        solve_mpqp(object(), algorithm=mpqp_algorithm.geometric)      # the geometric driver needs an mpQP
    assert {a.value for a in mpqp_algorithm} >= {'combinatorial', 'p combinatorial', 'p combinatorial exp', 'graph'}
    assert 'mpqp_algorithm.combinatorial' in mpqp_algorithm.all_algos()


def test_mask_round_trip():
    from ppopt_amd._lib import masks_to_sets, sets_to_masks
    sets = [(0,), (5, 63, 64), (127,), (1, 2, 3, 70, 100)]
    assert masks_to_sets(sets_to_masks(sets)) == sets


def test_region_batch_lazy_fields():
    """BatchCriticalRegion cuts its fields out of the compact arrays of include/mpcombi.h on first access."""
    from ppopt_amd.region_batch import RegionBatch
    n_x, n_t, n_c, n_tc, k = 3, 2, 5, 4, 2
    fd = n_x * n_t + n_x + k * n_t + k
    fi = 8 + k + n_tc + k + 2 * (n_c - k)
    hd = numpy.arange(2 * fd, dtype=float).reshape(2, fd)
    hi = -numpy.ones((2, fi), dtype=numpy.int32)
    hi[0, :8] = [3, 7, 2, 1, 1, 2, 0, 0]; hi[1, :8] = [3, 9, 1, 0, 2, 0, 2, 0]
    hi[0, 8:10] = [1, 4]; hi[0, 10] = 2; hi[0, 14] = 4; hi[0, 16:18] = [0, 2]; hi[0, 19:21] = [0, 3]
    hi[1, 8:10] = [0, 2]; hi[1, 14:16] = [0, 2]
    er = numpy.arange(9, dtype=float).reshape(3, 3)
    regs = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, k).regions()
    r0, r1 = regs
    assert isinstance(r0, CriticalRegion) and r0.active_set == [1, 4] and r1.active_set == [0, 2]
    assert r0.A.shape == (3, 2) and r0.A[0, 1] == 1.0 and r0.b.shape == (3, 1) and r0.b[0, 0] == 6.0
    assert r0.C.shape == (2, 2) and r0.d.shape == (2, 1) and r1.A[0, 0] == fd
    assert r0.E.shape == (2, 2) and numpy.array_equal(r0.f.ravel(), [0.0, 3.0]) and numpy.array_equal(r0.E[1], [4.0, 5.0])
    assert r1.E.shape == (1, 2) and r1.f[0, 0] == 6.0
    assert r0.omega_set == [2] and r0.lambda_set == [4] and r0.regular_set == [[0, 2], [0, 3]]
    assert r1.omega_set == [] and r1.lambda_set == [0, 2] and r1.regular_set == [[], []]
    r0.A = numpy.zeros((3, 2))            # assignable like a dataclass field
    assert numpy.all(r0.A == 0) and 'active set [1, 4]' in repr(r0)
    th = numpy.ones((2, 1))
    assert r1.evaluate(th).shape == (3, 1) and r1.materialize() is r1


def test_solution_stacks_regions_for_the_locator():
    """Solution._stacked (input of mpc_locator_create): [f | E] rows, row offsets and [b | A] of every region in list
    order -- cut out of the level arrays for batch-backed regions, field by field for plain CriticalRegion objects."""
    from ppopt_amd.region_batch import RegionBatch
    from ppopt_amd.solution import Solution
    rng = numpy.random.default_rng(1)
    n_x, n_t, n_c, n_tc, k = 3, 2, 5, 4, 2
    fd = n_x * n_t + n_x + k * n_t + k
    fi = 8 + k + n_tc + k + 2 * (n_c - k)
    hd = rng.random((4, fd))
    hi = -numpy.ones((4, fi), dtype=numpy.int32)
    er = rng.random((9, n_t + 1))
    # slots 0, 2, 3 are regions owning rows [5:8], [0:2], [2:5] of the pool (not in slot order); slot 1 is not a region
    for j, (st, nE, off) in enumerate([(3, 3, 5), (2, 0, 0), (3, 2, 0), (3, 3, 2)]):
        hi[j, :8] = [st, j, nE, 0, 0, 0, off, 0]
        hi[j, 8:10] = [0, 1]
    batch_regs = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, k, numpy.array([0, 2, 3])).regions()
    plain = CriticalRegion(rng.random((n_x, n_t)), rng.random((n_x, 1)), rng.random((k, n_t)), rng.random((k, 1)),
                           rng.random((4, n_t)), rng.random((4, 1)), [0, 1], [], [], [[], []])

    class P:
        def num_t(self):
            return n_t
    sol = Solution(P(), [batch_regs[0], plain, batch_regs[1], batch_regs[2]])
    ef, row_off, xlaw = sol._stacked()
    assert row_off.tolist() == [0, 3, 7, 9, 12] and ef.shape == (12, n_t + 1) and xlaw.shape == (4, n_x, n_t + 1)
    for i, r in enumerate(sol.critical_regions):
        rows = ef[row_off[i]:row_off[i + 1]]
        assert numpy.array_equal(rows[:, :1], r.f) and numpy.array_equal(rows[:, 1:], r.E)
        assert numpy.array_equal(xlaw[i][:, :1], r.b) and numpy.array_equal(xlaw[i][:, 1:], r.A)


def test_graph_traversal_bookkeeping_on_the_host():
    """The mask helpers of the connected-graph traversals (mpqp_hip_combi_graph: the host form of what csrc/graph.hpp does on the
    device): neighbour rules of mpqp_combi_graph.py:88-143 and mpqp_graph.py:69-108, the visited-set book with hashed keys
    confirmed on the full mask (and its exact mode after a forced collision)."""
    import numpy
    from ppopt_amd.mp_solvers import mpqp_hip_combi_graph as G
    sets = [[0, 5, 70], [1, 2, 3], [0, 5, 70]]
    m = G._sets_to_masks(sets, 2)
    assert G._popcount(m).tolist() == [3, 3, 3]
    assert G._masks_to_index_rows(m, 3, 100).tolist() == [[0, 5, 70], [1, 2, 3], [0, 5, 70]]
    bit, eq = G._bit_table(100, 2), G._sets_to_masks([[0]], 2)[0]
    # combinatorial_graph: rank deficient {0,5,70} -> subsets without dropping the equality row 0; region {1,2,3} -> 3 subsets + 97 supersets
    nb = G._neighbours(m[:2], numpy.array([True, True]), numpy.array([False, True]), bit, eq)
    got = sorted(tuple(numpy.flatnonzero(numpy.unpackbits(r.view(numpy.uint8), bitorder='little'))) for r in nb)
    want = sorted([(0, 70), (0, 5)] + [(2, 3), (1, 3), (1, 2)] + [tuple(sorted({1, 2, 3, j})) for j in range(100) if j not in (1, 2, 3)])
    assert got == want
    # graph: supersets only through the facet constraints
    facets = numpy.zeros((2, 100), dtype=bool)
    facets[1, [7, 9]] = True
    nb = G._neighbours(m[:2], numpy.array([False, False]), numpy.array([False, True]), bit, eq, facets)
    assert sorted(G._popcount(nb).tolist()) == [4, 4] and len(nb) == 2
    book = G._SetBook(2)
    assert len(book.add(m)) == 2 and len(book.add(G._sets_to_masks([[1, 2, 3], [4]], 2))) == 1 and len(book.h) == 3
    rng = numpy.random.default_rng(0)
    big = rng.integers(0, 2 ** 62, size=(50000, 2), dtype=numpy.int64).astype(numpy.uint64)
    assert len(book.add(big)) == 50000 and len(book.add(big[:1000])) == 0
    # a hash collision (forced: every mask hashes to 0) switches the book to exact keys without losing or inventing a set
    b2 = G._SetBook(2)
    b2._hash = lambda masks: numpy.zeros(len(masks), dtype=numpy.uint64)
    first = b2.add(G._sets_to_masks([[1], [2]], 2))
    assert b2.exact and len(first) == 2
    assert len(b2.add(G._sets_to_masks([[2], [3]], 2))) == 1


def test_geometric_sub_active_set():
    """solver_utils.py:169-202: a full-rank subset of an overdetermined active set, equalities first."""
    import numpy
    from ppopt_amd.mp_solvers.mpqp_hip_geometric import _sub_active_set

    class P:
        A = numpy.array([[1.0, 0.0], [2.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
        equality_indices = []

        def num_x(self):
            return 2
    assert _sub_active_set(P(), [0, 1, 2, 3]) == [0, 2]
    P.equality_indices = [3]
    assert _sub_active_set(P(), [3, 0, 1, 2]) == [3, 0]


def test_batch_materialisation_equals_lazy_access():
    """Solution.materialize / region_batch.materialize_regions: the fields cut out batch-wise are the fields each region would
    cut out lazily; fields that were assigned before are kept; plain CriticalRegion objects are left alone."""
    from ppopt_amd.region_batch import RegionBatch
    from ppopt_amd.solution import Solution
    rng = numpy.random.default_rng(2)
    n_x, n_t, n_c, n_tc, k = 4, 3, 9, 6, 3
    fd = n_x * n_t + n_x + k * n_t + k
    fi = 8 + k + n_tc + k + 2 * (n_c - k)
    n = 7
    hd = rng.random((n, fd))
    hi = -numpy.ones((n, fi), dtype=numpy.int32)
    er = rng.random((40, n_t + 1))
    off = 0
    for j in range(n):
        nE, n_om, n_la, n_re = int(rng.integers(1, 6)), int(rng.integers(0, n_tc + 1)), int(rng.integers(0, k + 1)), int(rng.integers(0, n_c - k + 1))
        hi[j, :8] = [3, j, nE, n_om, n_la, n_re, off, 0]
        off += nE
        hi[j, 8:8 + k] = sorted(rng.choice(n_c, k, replace=False).tolist())
        hi[j, 8 + k:8 + k + n_om] = numpy.arange(n_om)
        hi[j, 8 + k + n_tc:8 + k + n_tc + n_la] = numpy.arange(n_la)
        hi[j, 8 + 2 * k + n_tc:8 + 2 * k + n_tc + n_re] = numpy.arange(n_re)
        hi[j, 8 + 2 * k + n_tc + (n_c - k):8 + 2 * k + n_tc + (n_c - k) + n_re] = 10 + numpy.arange(n_re)
    slots = numpy.array([0, 2, 3, 5, 6])
    lazy = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, k, slots).regions()
    eager = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, k, slots).regions()
    marker = numpy.zeros((n_x, n_t))
    eager[1].A = marker
    plain = CriticalRegion(numpy.ones((n_x, n_t)), numpy.ones((n_x, 1)), numpy.ones((k, n_t)), numpy.ones((k, 1)), numpy.ones((2, n_t)),
                           numpy.ones((2, 1)), [0, 1, 2], [], [], [[], []])

    class P:
        def num_t(self):
            return n_t
    sol = Solution(P(), eager + [plain])
    assert sol.materialize() is sol
    assert eager[1].A is marker
    for a, b in zip(lazy, eager):
        for fld in ('b', 'C', 'd', 'E', 'f') + (() if b is eager[1] else ('A',)):
            assert fld in b.__dict__ and numpy.array_equal(getattr(a, fld), getattr(b, fld)), fld
        assert a.active_set == b.active_set and a.omega_set == b.omega_set and a.lambda_set == b.lambda_set and a.regular_set == b.regular_set


def test_gc_paused_is_reentrant_and_restores_the_collector(monkeypatch):
    """region_batch.gc_paused (used by solve_many): the cycle collector is held while the region objects are created -- nested solves and solves on other
    threads share one pause, the collector's previous state comes back with the outermost exit (also after an exception, also when it
    was disabled to begin with), and MPC_KEEP_GC=1 leaves it alone."""
    import gc
    import threading
    from ppopt_amd.region_batch import gc_paused
    monkeypatch.delenv('MPC_KEEP_GC', raising=False)
    assert gc.isenabled()
    with gc_paused():
        assert not gc.isenabled()
        with gc_paused():
            assert not gc.isenabled()
        assert not gc.isenabled()
        seen = []

        def other():
            with gc_paused():
                seen.append(gc.isenabled())
        t = threading.Thread(target=other); t.start(); t.join()
        assert seen == [False] and not gc.isenabled()
    assert gc.isenabled()
    try:
        with gc_paused():
            raise RuntimeError('x')
    except RuntimeError:
        pass
    assert gc.isenabled()
    gc.disable()
    try:
        with gc_paused():
            assert not gc.isenabled()
        assert not gc.isenabled()      # it was off before: it stays off
    finally:
        gc.enable()
    monkeypatch.setenv('MPC_KEEP_GC', '1')
    with gc_paused():
        assert gc.isenabled()


def test_gc_pause_hands_the_new_objects_to_the_oldest_generation_without_losing_any_garbage(monkeypatch):
    """Round 6 (region_batch._promote_young; solve() and solve_distributed() hold the collector now): at the end of the
    outermost pause the young generation is spliced into the oldest one (gc.freeze(); gc.unfreeze()) instead of being walked by the next
    allocation -- the collector comes back enabled with an empty young generation, nothing stays frozen, a reference cycle that became
    garbage during the pause is still found by the next full collection, objects the caller has frozen deliberately stay frozen, and
    MPC_GC_PROMOTE=0 / a short pause leave the generations alone."""
    import gc
    import weakref
    from ppopt_amd.region_batch import gc_paused
    monkeypatch.delenv('MPC_KEEP_GC', raising=False)
    monkeypatch.delenv('MPC_GC_PROMOTE', raising=False)

    class Node:
        pass
    gc.collect()
    assert gc.isenabled() and gc.get_freeze_count() == 0
    with gc_paused():
        keep = [[i] for i in range(5000)]
        a, b = Node(), Node()
        a.other, b.other = b, a
        ref = weakref.ref(a)
        del a, b                      # a garbage cycle made while the collector is held
    assert gc.isenabled() and gc.get_freeze_count() == 0
    assert gc.get_count()[0] < 100    # (no young objects left for the next allocation to walk)
    assert ref() is not None
    gc.collect()
    assert ref() is None and len(keep) == 5000
    # short pause: nothing to promote
    gc.collect()
    before = gc.get_count()
    with gc_paused():
        few = [[i] for i in range(10)]
    assert gc.isenabled() and gc.get_count()[0] >= before[0] and len(few) == 10
    # the switch
    calls = []
    real_freeze = gc.freeze
    monkeypatch.setattr(gc, 'freeze', lambda: (calls.append(1), real_freeze())[1])
    monkeypatch.setenv('MPC_GC_PROMOTE', '0')
    with gc_paused():
        many = [[i] for i in range(5000)]
    assert gc.isenabled() and not calls and len(many) == 5000
    monkeypatch.delenv('MPC_GC_PROMOTE')
    with gc_paused():
        many2 = [[i] for i in range(5000)]
    assert gc.isenabled() and len(calls) == 1 and gc.get_freeze_count() == 0 and len(many2) == 5000
    monkeypatch.setattr(gc, 'freeze', real_freeze)
    # a caller's own frozen objects are not thawed
    gc.collect()
    gc.freeze()
    try:
        frozen = gc.get_freeze_count()
        assert frozen > 0
        with gc_paused():
            more = [[i] for i in range(5000)]
        assert gc.get_freeze_count() == frozen and gc.isenabled() and len(more) == 5000
    finally:
        gc.unfreeze()


def test_materialisation_by_the_c_loop_equals_the_python_loop(monkeypatch):
    """Round 6: Solution.materialize runs its per-region loop in C when ppopt_amd/_fastmat.so is built (csrc_host/fastmat.c, __graft_entry__.build)
    and in Python otherwise: the same fields with the same values, shapes and element types either way -- for slots in order, for a
    selection of slots out of order, with a field assigned before (kept), and for regions without rows / with empty index lists."""
    import ppopt_amd.region_batch as rb
    from ppopt_amd.region_batch import RegionBatch
    if rb._fastmat is None:
        pytest.skip('ppopt_amd/_fastmat.so is not built (python __graft_entry__.py builds it)')
    rng = numpy.random.default_rng(5)
    n_x, n_t, n_c, n_tc, k = 5, 3, 11, 6, 2
    fd = n_x * n_t + n_x + k * n_t + k
    fi = 8 + k + n_tc + k + 2 * (n_c - k)
    n = 40
    hd = rng.random((n, fd))
    hi = -numpy.ones((n, fi), dtype=numpy.int32)
    off = 0
    for j in range(n):
        nE = int(rng.integers(0, 6))
        n_om, n_la, n_re = int(rng.integers(0, n_tc + 1)), int(rng.integers(0, k + 1)), int(rng.integers(0, n_c - k + 1))
        hi[j, :8] = [3, j, nE, n_om, n_la, n_re, off, 0]
        off += nE
        hi[j, 8:8 + k] = sorted(rng.choice(n_c, k, replace=False).tolist())
        hi[j, 8 + k:8 + k + n_om] = rng.integers(0, n_tc, n_om)
        hi[j, 8 + k + n_tc:8 + k + n_tc + n_la] = rng.integers(0, k, n_la)
        hi[j, 8 + 2 * k + n_tc:8 + 2 * k + n_tc + n_re] = rng.integers(0, n_c, n_re)
        hi[j, 8 + 2 * k + n_tc + (n_c - k):8 + 2 * k + n_tc + (n_c - k) + n_re] = rng.integers(0, n_c, n_re)
    er = rng.random((off, n_t + 1))
    for slots in (numpy.arange(n), numpy.array([7, 3, 21, 22, 23, 39, 0])):
        by_c = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, k, slots).regions()
        by_py = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, k, slots).regions()
        marker = numpy.zeros((n_x, 1))
        by_c[2].b = marker
        by_py[2].b = marker
        rb.materialize_regions(by_c)
        monkeypatch.setattr(rb, '_fastmat', None)
        rb.materialize_regions(by_py)
        monkeypatch.undo()
        assert by_c[2].b is marker and by_py[2].b is marker
        for a, b in zip(by_c, by_py):
            assert set(a.__dict__) == set(b.__dict__) == set(rb._FIELD_NAMES)
            for fld in ('A', 'b', 'C', 'd', 'E', 'f'):
                x, y = getattr(a, fld), getattr(b, fld)
                assert x.shape == y.shape and x.dtype == y.dtype and numpy.array_equal(x, y), fld
                assert x.size == 0 or numpy.shares_memory(x, hd) or numpy.shares_memory(x, er) or x is marker
            assert a.active_set == b.active_set and a.omega_set == b.omega_set and a.lambda_set == b.lambda_set and a.regular_set == b.regular_set
            assert all(type(v) is int for v in a.active_set + a.omega_set + a.lambda_set + a.regular_set[0] + a.regular_set[1])
    # a header that does not fit its arrays: the C loop declines, the Python loop takes over (here it raises nothing: slices clip)
    bad = hi.copy()
    bad[5, 2] = 10 ** 6
    regs = RegionBatch(hd, bad, er, n_x, n_t, n_c, n_tc, k).regions()
    rb.materialize_regions(regs)
    assert regs[4].E.shape[1] == n_t and 'A' in regs[5].__dict__

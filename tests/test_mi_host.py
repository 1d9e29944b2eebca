"""Host logic of the mixed-integer caller against goldens captured from the reference
(oracle/ref_harness/gen_mi_goldens.py); no device needed."""
import glob
import os

import numpy
import pytest

from ppopt_amd.critical_region import CriticalRegion
from ppopt_amd.solver import Solver
from ppopt_amd.utils.region_overlap_utils import get_bounds_1d, reduce_overlapping_critical_regions_1d

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MI_FILES = sorted(glob.glob(os.path.join(GOLDEN, 'mi_*.npz')))


def unpack_regions(g, prefix, with_y=True):
    regs = []
    for i in range(int(g[prefix + 'n'])):
        k, ne = int(g[prefix + 'k'][i]), int(g[prefix + 'nE'][i])
        r = CriticalRegion(g[prefix + 'A'][i].copy(), g[prefix + 'b'][i].reshape(-1, 1).copy(),
                           g[prefix + 'C'][i, :k].copy(), g[prefix + 'd'][i, :k].reshape(-1, 1).copy(),
                           g[prefix + 'E'][i, :ne].copy(), g[prefix + 'f'][i, :ne].reshape(-1, 1).copy(),
                           g[prefix + 'as'][i, :k].tolist())
        if with_y:
            r.y_fixation = g[prefix + 'y'][i].tolist()
            r.y_indices = g['binary_indices'].tolist()
            r.x_indices = g['cont_indices'].tolist()
        regs.append(r)
    return regs


class _ObjectiveOnly:
    """The part of an MPMILP_Program the overlap reduction touches, built from the golden's presolved matrices."""

    def __init__(self, g):
        self.c, self.H, self.c_c, self.c_t, self.Q_t = g['proc_c'], g['proc_H'], g['proc_c_c'], g['proc_c_t'], g['proc_Q_t']
        self._nt = g['proc_F'].shape[1]

    def num_t(self):
        return self._nt

    def evaluate_objective(self, x, th):
        v = th.T @ self.H.T @ x + self.c.T @ x + self.c_c + self.c_t.T @ th + 0.5 * th.T @ self.Q_t @ th
        return float(v[0, 0])


def test_goldens_present():
    assert len(MI_FILES) == 13


def test_binary_fixations_order_is_tree_leaf_order():
    fix = Solver.binary_fixations(3)
    assert fix.tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]]
    assert Solver.binary_fixations(0).shape == (1, 0)
    for path in MI_FILES:
        g = numpy.load(path)
        nb = len(g['binary_indices'])
        order = {tuple(r): i for i, r in enumerate(Solver.binary_fixations(nb).tolist())}
        ranks = [order[tuple(c)] for c in g['combos'].tolist()]
        assert ranks == sorted(ranks), path


@pytest.mark.parametrize('path', MI_FILES, ids=[os.path.basename(p)[3:-4] for p in MI_FILES])
def test_overlap_reduction_matches_reference(path):
    """P_ (enumeration output of the reference) -> our reduction -> F_ (the reference's final list), in list order."""
    g = numpy.load(path)
    reducible = g['proc_F'].shape[1] == 1 and 'proc_Q' not in g.files
    if not reducible:
        assert int(g['P_n']) == int(g['F_n'])
        return
    regions, still = reduce_overlapping_critical_regions_1d(_ObjectiveOnly(g), unpack_regions(g, 'P_'))
    want = unpack_regions(g, 'F_')
    assert still == bool(g['F_overlapping'])
    assert len(regions) == len(want)
    for got, ref in zip(regions, want):
        assert got.y_fixation == ref.y_fixation and got.active_set == ref.active_set
        numpy.testing.assert_allclose(get_bounds_1d(got.E, got.f), get_bounds_1d(ref.E, ref.f), rtol=0, atol=1e-9)
        numpy.testing.assert_allclose(got.A, ref.A, atol=1e-12)
        numpy.testing.assert_allclose(got.b, ref.b, atol=1e-12)


def test_reduction_rejects_more_than_one_parameter():
    g = numpy.load(os.path.join(GOLDEN, 'mi_acevedo_mpmilp.npz'))
    with pytest.raises(ValueError):
        reduce_overlapping_critical_regions_1d(_ObjectiveOnly(g), [])


def test_mitree_from_leaf_table():
    """Tree structure from a table of leaf verdicts: node count and leaf order of the reference's goldens."""
    from ppopt_amd.mp_solvers.mitree import MITree

    class Fake:
        def __init__(self, nb, feasible):
            self.binary_indices = list(range(nb))
            self._ok = set(map(tuple, feasible))

        def check_bin_feasibility(self, fix):
            return any(c[:len(fix)] == tuple(fix) for c in self._ok)

    for path in MI_FILES:
        g = numpy.load(path)
        nb = len(g['binary_indices'])
        tree = MITree(Fake(nb, g['combos'].tolist()))
        assert tree.count_nodes() == int(g['n_nodes']), path
        assert [leaf.fixed_bins for leaf in tree.get_full_leafs()] == g['combos'].tolist(), path


# ---- bounded-footprint MILP batches (Solver.milp_leaf_feasibility / milp_any_feasible / solve_milp) ------------------------
def _scipy_lp_batch(counter):
    """Stand-in for the device LP batch (ppopt_amd._lib.lp_solve_batch) so that the HOST logic -- chunking, relaxation
    pruning, early exit per row -- is testable without a GPU.  Checker only; the product path has no CPU solver."""
    from scipy.optimize import linprog

    def lp_solve_batch(A, b, c, eq_flags, device=0, want_x=True):
        n_lp, m = eq_flags.shape
        n = A.shape[-1]
        counter['lps'] += n_lp
        counter['calls'] = counter.get('calls', 0) + 1
        counter['max_flag_bytes'] = max(counter['max_flag_bytes'], eq_flags.nbytes)
        status = numpy.zeros(n_lp, dtype=numpy.int32)
        x = numpy.zeros((n_lp, n)) if want_x else None
        obj = numpy.zeros(n_lp)
        for i in range(n_lp):
            eq = eq_flags[i].astype(bool)
            Ai = A[i] if A.ndim == 3 else A                      # per-instance or shared data, as mpc_lp_solve_batch takes it
            bi = b[i] if numpy.ndim(b) == 2 else b
            ci = None if c is None else (c[i] if numpy.ndim(c) == 2 else c)
            res = linprog(numpy.zeros(n) if ci is None else ci, A_ub=Ai[~eq], b_ub=bi[~eq], A_eq=Ai[eq] if eq.any() else None,
                          b_eq=bi[eq] if eq.any() else None, bounds=(None, None), method='highs-ds')
            status[i] = 0 if res.status == 0 else (2 if res.status == 3 else 1)
            if res.status == 0:
                obj[i] = res.fun
                if want_x:
                    x[i] = res.x
        return status, x, obj, numpy.zeros(n_lp, dtype=numpy.int32)
    return lp_solve_batch


def _cardinality_milp(nb, max_ones):
    """x in R^1, y in {0,1}^nb:  sum(y) <= max_ones,  x <= sum_j 2^-j y_j,  -x <= 0   (every leaf with few ones is feasible)."""
    n = 1 + nb
    A = numpy.zeros((3, n))
    A[0, 1:] = 1.0
    A[1, 0] = 1.0
    A[1, 1:] = -(0.5 ** numpy.arange(nb))
    A[2, 0] = -1.0
    b = numpy.array([float(max_ones), 0.0, 0.0])
    return A, b, list(range(1, n))


def test_milp_leaf_feasibility_prunes_through_relaxations(monkeypatch):
    from ppopt_amd import solver as solver_mod
    counter = {'lps': 0, 'max_flag_bytes': 0}
    monkeypatch.setattr(solver_mod._lib, 'lp_solve_batch', _scipy_lp_batch(counter))
    nb = 9
    A, b, bins = _cardinality_milp(nb, 2)
    S = Solver()
    S.MILP_DIRECT_BINARIES = 3          # walk the tree from depth 3 on
    S.MILP_BATCH_BYTES = 2048           # and force many small device batches
    table = S.milp_leaf_feasibility(A, b, [], bins)
    ones = numpy.array([bin(i).count('1') for i in range(1 << nb)])
    assert numpy.array_equal(table, ones <= 2)
    assert counter['lps'] < (1 << nb)                   # fewer LPs than leaves: whole subtrees were never posed
    assert counter['max_flag_bytes'] <= 2048
    # the direct form (few binaries) gives the same table
    S2 = Solver()
    assert numpy.array_equal(S2.milp_leaf_feasibility(A, b, [], bins), table)


def test_milp_any_feasible_and_solve_milp_with_bounded_batches(monkeypatch):
    from ppopt_amd import solver as solver_mod
    counter = {'lps': 0, 'max_flag_bytes': 0}
    monkeypatch.setattr(solver_mod._lib, 'lp_solve_batch', _scipy_lp_batch(counter))
    nb = 5
    A, b, bins = _cardinality_milp(nb, 2)
    S = Solver()
    S.MILP_BATCH_BYTES = 512
    leaves = numpy.flatnonzero(S.milp_leaf_feasibility(A, b, [], bins))
    # row 0 tight: sum(y) == 2 has leaves; row 1 tight: x == sum 2^-j y_j fine; row 2 tight: x == 0 fine; rows {0} with max_ones
    got = S.milp_any_feasible(A, b, [[0], [1], [2], [1, 2]], bins, leaves)
    assert got.tolist() == [True, True, True, True]
    A2 = numpy.vstack([A, [[1.0] + [0.0] * nb]])          # x <= -1 can never hold as an equality (x >= 0)
    b2 = numpy.concatenate([b, [-1.0]])
    assert S.milp_any_feasible(A2, b2, [[3], [0]], bins).tolist() == [False, False]   # the system itself is infeasible
    b2[-1] = 5.0                                          # x <= 5: never tight (x <= sum 2^-j y_j < 2)
    lv2 = numpy.flatnonzero(S.milp_leaf_feasibility(A2, b2, [], bins))
    assert S.milp_any_feasible(A2, b2, [[3], [0]], bins, lv2).tolist() == [False, True]
    assert counter['max_flag_bytes'] <= 512
    # solve_milp: min -x -> x = largest sum of two weights = 1 + 1/2, y = (1, 1, 0, 0, 0); first fixation wins ties
    c = numpy.zeros(1 + nb)
    c[0] = -1.0
    out = S.solve_milp(c, A, b, [], bins)
    assert abs(out.obj + 1.5) <= 1e-9 and out.sol[1:].tolist() == [1, 1, 0, 0, 0] and abs(out.sol[0] - 1.5) <= 1e-9
    assert S.solve_milp(c, A2, numpy.concatenate([b, [-1.0]]), [], bins) is None
    # dense form agrees on status, and refuses outputs beyond its footprint instead of exhausting the host
    st, x, obj = S.solve_milp_batch(None, A, b, [[], [0]], bins)
    ones = numpy.array([bin(i).count('1') for i in range(1 << nb)])
    assert numpy.array_equal(st[0] == 0, ones <= 2) and numpy.array_equal(st[1] == 0, ones == 2)
    assert x.shape == (2, 1 << nb, 1 + nb) and numpy.array_equal(x[0][:, 1:], Solver.binary_fixations(nb))
    S.MILP_BATCH_BYTES = 16
    with pytest.raises(MemoryError):
        S.solve_milp_batch(None, A, b, [[]] * 64, bins)


# ---- LP calls of several threads posed as shared device batches (solver.LPCoalescer; the presolve of the enumeration's sub-programs) ----
def test_lp_coalescer_poses_the_calls_of_all_threads_together(monkeypatch):
    from concurrent.futures import ThreadPoolExecutor
    from ppopt_amd import solver as solver_mod
    from ppopt_amd.solver import LPCoalescer
    counter = {'lps': 0, 'max_flag_bytes': 0}
    monkeypatch.setattr(solver_mod._lib, 'lp_solve_batch', _scipy_lp_batch(counter))
    rng = numpy.random.default_rng(5)

    def box_lp(n, shift):      # a box around `shift`, two extra cuts; feasible, bounded
        A = numpy.vstack([numpy.eye(n), -numpy.eye(n), rng.standard_normal((2, n))])
        b = numpy.concatenate([shift + 1.0, 1.0 - shift, [5.0, 5.0]])
        return A, b, rng.standard_normal(n)
    jobs = [box_lp(3 if j % 2 else 4, rng.standard_normal(3 if j % 2 else 4) * 0.1) for j in range(10)]

    def work(solver, job, extra_round):
        A, b, c = job
        out = [solver.solve_lp(c, A, b, [])]                                           # stage 1: one LP
        out.append(solver.solve_lp_batch(None, A, b, [[i] for i in range(A.shape[0])]))    # stage 2: one LP per row
        if extra_round:
            out.append(solver.solve_lp(-c, A, b, [0]))                                 # some workers make a third call
        return out
    plain = Solver()
    want = [work(plain, job, j % 3 == 0) for j, job in enumerate(jobs)]
    calls_plain = counter['calls']
    counter['calls'] = 0
    co = LPCoalescer(plain, len(jobs))
    parked = co.solver()

    def worker(args):
        j, job = args
        try:
            return work(parked, job, j % 3 == 0)
        finally:
            co.worker_done()
    with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
        got = list(pool.map(worker, enumerate(jobs)))
    assert counter['calls'] <= 6 < calls_plain            # three stages x two shapes at most, against 24 separate calls
    assert co.n_calls == calls_plain

    def same(a, b):
        if a is None or b is None:
            return a is None and b is None
        return a.obj == b.obj and numpy.array_equal(a.sol, b.sol) and numpy.array_equal(a.active_set, b.active_set)
    for w, g in zip(want, got):
        assert same(w[0], g[0]) and len(w[1]) == len(g[1]) and all(same(x, y) for x, y in zip(w[1], g[1]))
        assert len(w) == len(g) and (len(w) == 2 or same(w[2], g[2]))


def test_lp_coalescer_hands_a_failure_to_every_parked_caller(monkeypatch):
    from concurrent.futures import ThreadPoolExecutor
    from ppopt_amd import solver as solver_mod
    from ppopt_amd.solver import LPCoalescer

    def broken(*a, **k):
        raise RuntimeError('device lost')
    monkeypatch.setattr(solver_mod._lib, 'lp_solve_batch', broken)
    co = LPCoalescer(Solver(), 4)
    parked = co.solver()
    A, b = numpy.vstack([numpy.eye(2), -numpy.eye(2)]), numpy.ones(4)

    def worker(_):
        try:
            parked.solve_lp(None, A, b, [])
            return 'no error'
        except RuntimeError as ex:
            return str(ex)
        finally:
            co.worker_done()
    with ThreadPoolExecutor(max_workers=4) as pool:
        assert list(pool.map(worker, range(4))) == ['device lost'] * 4


def test_substituted_rows_are_cached_per_program_and_follow_row_changes():
    """MPMILP_Program._substituted_rows finds the rows with continuous / parametric content once per program; a program whose rows
    are replaced afterwards must not reuse the old selection."""
    import warnings
    from ppopt_amd import MPMILP_Program
    A = numpy.array([[1.0, 0.0, 1.0], [0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, -1.0, 0.0]])
    b = numpy.array([[2.0], [1.0], [0.0], [1.0], [0.0]])
    F = numpy.array([[1.0], [0.0], [0.0], [0.0], [0.0]])

    class NoLP(Solver):         # the constructor's presolve is not what is tested here
        def solve_lp_batch(self, c, A, b, equality_sets):
            return [object()] * len(equality_sets)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = MPMILP_Program(A, b, numpy.zeros((3, 1)), numpy.zeros((3, 1)), numpy.array([[1.0], [-1.0]]), numpy.array([[1.0], [1.0]]), F,
                              binary_indices=[2], solver=NoLP(), post_process=False)
    r0 = prog._substituted_rows([0])
    r1 = prog._substituted_rows([1])
    assert numpy.array_equal(r0[0], r1[0]) and not numpy.array_equal(r0[1], r1[1])      # same rows, right-hand side moved by A_bin y
    kept_before = r0[0].shape[0]
    prog.A = prog.A[:-1].copy(); prog.b = prog.b[:-1].copy(); prog.F = prog.F[:-1].copy()
    assert prog._substituted_rows([0])[0].shape[0] == kept_before - 1


def test_closing_rows_of_a_parameter_set_without_a_vertex(monkeypatch):
    """MPLP_Program._engine_parameter_rows (host logic, LPs through the stand-in): a parameter set with a vertex is handed over as it
    is; a slab gets the widened box of theta over the whole program appended (2 n_theta rows, none of them reachable); a box of big-M size or an
    unbounded direction leaves the rows alone."""
    import warnings
    from ppopt_amd import MPQP_Program, solver as solver_mod
    counter = {'lps': 0, 'max_flag_bytes': 0}
    monkeypatch.setattr(solver_mod._lib, 'lp_solve_batch', _scipy_lp_batch(counter))

    def program(x_box, theta_rows):
        # x in R^2, theta in R^2:  |x_i| <= x_box,  x_1 - theta_1 = 0 as two inequalities (theta_1 bounded through x),  theta_2 rows given
        A = numpy.array([[1.0, 0], [-1, 0], [0, 1], [0, -1], [1, 0], [-1, 0]])
        b = numpy.array([[x_box], [x_box], [x_box], [x_box], [0.0], [0.0]])
        F = numpy.array([[0.0, 0], [0, 0], [0, 0], [0, 0], [1, 0], [-1, 0]])
        A_t = numpy.array(theta_rows, dtype=float).reshape(-1, 2)
        b_t = numpy.ones((A_t.shape[0], 1)) * 3.0
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return MPQP_Program(A, b, numpy.zeros((2, 1)), numpy.zeros((2, 2)), numpy.eye(2), A_t, b_t, F, post_process=False)
    with_vertex = program(2.0, [[1, 0], [-1, 0], [0, 1], [0, -1]])
    A_t, b_t = with_vertex._engine_parameter_rows()
    assert A_t is with_vertex.A_t and b_t is with_vertex.b_t
    slab = program(2.0, [[0, 1], [0, -1]])          # theta_1 is bounded only through x_1 = theta_1, |x_1| <= 2
    A_t, b_t = slab._engine_parameter_rows()
    assert A_t.shape == (2 + 4, 2) and numpy.array_equal(A_t[:2], slab.A_t)
    lo, hi = -(b_t[4:, 0]), b_t[2:4, 0]             # closing rows: theta <= hi + pad, -theta <= -(lo - pad)
    scale = numpy.linalg.norm([1.0, 1.0])           # the constructor scales [A | -F] rows to unit norm: x_1 - theta_1 = 0 keeps theta_1 = x_1
    assert numpy.all(hi > numpy.array([2.0, 3.0]) + 0.9) and numpy.all(lo < -numpy.array([2.0, 3.0]) - 0.9), (lo, hi, scale)
    assert slab._engine_parameter_rows()[0] is A_t  # decided once per set of rows
    big = program(1e7, [[0, 1], [0, -1]])
    assert big._engine_parameter_rows()[0] is big.A_t
    open_set = program(2.0, [[0, 1]])               # theta_2 unbounded below
    assert open_set._engine_parameter_rows()[0] is open_set.A_t
    monkeypatch.setenv('MPC_NO_THETA_CLOSE', '1')
    assert slab._engine_parameter_rows()[0] is slab.A_t

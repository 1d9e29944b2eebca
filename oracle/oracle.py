"""ctypes front-end of the CPU oracle (oracle/mpcombi_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by ppopt_amd/.  Mirrors the reference's Python-level interfaces (Solver.solve_lp,
program.check_feasibility/check_optimality, gen_cr_from_active_set, the parallel driver) on top
of the C restatement so that parity tests read like the reference's own tests.
"""
import ctypes
import os
import subprocess
from typing import Dict, List, Optional, Tuple

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libmpcombi_oracle.so')

INFEASIBLE, FEASIBLE, OPTIMAL_NO_REGION, REGION, SINGULAR_KKT, LP_LIMIT = range(6)

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int32_p = ctypes.POINTER(ctypes.c_int32)
_c_uint8_p = ctypes.POINTER(ctypes.c_uint8)


class _Problem(ctypes.Structure):
    _fields_ = [('n_x', ctypes.c_int32), ('n_t', ctypes.c_int32), ('n_c', ctypes.c_int32), ('n_eq', ctypes.c_int32),
                ('n_tc', ctypes.c_int32), ('is_qp', ctypes.c_int32), ('A', _c_double_p), ('b', _c_double_p),
                ('F', _c_double_p), ('c', _c_double_p), ('H', _c_double_p), ('Q', _c_double_p), ('A_t', _c_double_p),
                ('b_t', _c_double_p)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'mpcombi_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'libmpcombi_oracle.so'])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_lp_solve.restype = ctypes.c_int
        L.orc_lp_solve.argtypes = [ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int,
                                   _c_int32_p, _c_double_p, _c_double_p, _c_int32_p]
        L.orc_set_feas_tol.restype = None
        L.orc_set_feas_tol.argtypes = [ctypes.c_double]
        L.orc_is_full_rank.restype = ctypes.c_int
        L.orc_is_full_rank.argtypes = [_c_double_p, ctypes.c_int, _c_int32_p, ctypes.c_int]
        L.orc_singular_values.restype = ctypes.c_int
        L.orc_singular_values.argtypes = [_c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p]
        pp = ctypes.POINTER(_Problem)
        L.orc_check_feasibility.restype = ctypes.c_int
        L.orc_check_feasibility.argtypes = [pp, _c_int32_p, ctypes.c_int, ctypes.c_int]
        L.orc_check_optimality.restype = ctypes.c_int
        L.orc_check_optimality.argtypes = [pp, _c_int32_p, ctypes.c_int]
        L.orc_optimal_control_law.restype = ctypes.c_int
        L.orc_optimal_control_law.argtypes = [pp, _c_int32_p, ctypes.c_int, _c_double_p, _c_double_p, _c_double_p,
                                              _c_double_p]
        L.orc_region_doubles.restype = ctypes.c_int64
        L.orc_region_doubles.argtypes = [pp]
        L.orc_region_ints.restype = ctypes.c_int64
        L.orc_region_ints.argtypes = [pp]
        L.orc_gen_cr.restype = ctypes.c_int
        L.orc_gen_cr.argtypes = [pp, _c_int32_p, ctypes.c_int, _c_double_p, _c_int32_p]
        L.orc_full_process.restype = ctypes.c_int
        L.orc_full_process.argtypes = [pp, _c_int32_p, ctypes.c_int, _c_double_p, _c_int32_p]
        L.orc_check_level.restype = ctypes.c_int
        L.orc_check_level.argtypes = [pp, _c_int32_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_uint8_p,
                                      _c_double_p, _c_int32_p]
        L.orc_generate_children.restype = ctypes.c_int64
        L.orc_generate_children.argtypes = [pp, _c_int32_p, ctypes.c_int64, ctypes.c_int, _c_uint8_p, _c_int32_p,
                                            _c_int32_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_int32_p,
                                            ctypes.c_int64]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(_c_double_p)


def _ip(a):
    return a.ctypes.data_as(_c_int32_p)


def lp_solve(c, A, b, equality_constraints=None) -> Tuple[int, Optional[numpy.ndarray], float, int]:
    """min c'x s.t. Ax <= b, rows `equality_constraints` as equalities.  Returns (status, x, obj, iterations)."""
    A = numpy.ascontiguousarray(A, dtype=numpy.float64)
    m, n = A.shape
    b = numpy.ascontiguousarray(b, dtype=numpy.float64).reshape(-1)
    eq = numpy.ascontiguousarray([] if equality_constraints is None else equality_constraints, dtype=numpy.int32)
    x = numpy.zeros(n)
    obj = ctypes.c_double(0.0)
    it = ctypes.c_int32(0)
    cp = None
    if c is not None:
        cc = numpy.ascontiguousarray(c, dtype=numpy.float64).reshape(-1)
        cp = _dp(cc)
    st = lib().orc_lp_solve(m, n, _dp(A), _dp(b), cp, len(eq), _ip(eq), _dp(x), ctypes.byref(obj), ctypes.byref(it))
    return st, (x if st == 0 else None), obj.value, it.value


def set_feas_tol(tol: float = 1e-7):
    """Primal feasibility tolerance of the oracle simplex (used to classify knife-edge decisions)."""
    lib().orc_set_feas_tol(float(tol))


def singular_values(M) -> numpy.ndarray:
    M = numpy.ascontiguousarray(M, dtype=numpy.float64)
    sv = numpy.zeros(max(M.shape))
    n = lib().orc_singular_values(_dp(M), M.shape[0], M.shape[1], _dp(sv))
    return sv[:n]


class OracleLPOutput:
    """The fields of SolverOutput (solver_interface_utils.py:7-40) the hot path consumes."""

    def __init__(self, obj, sol):
        self.obj = obj
        self.sol = sol


class OracleSolver:
    """Deterministic-solver plug (solver.py:211 `Solver.solve_lp`) backed by the C oracle; used by CPU tests to
    construct programs (presolve LPs) without a GPU."""
    solvers = {'lp': 'oracle'}

    def solve_lp(self, c, A, b, equality_constraints=None, verbose=False, get_duals=True):
        if A is None or A.shape[0] == 0 or A.shape[1] == 0:  # cvxopt_interface.py:186-190
            return None
        st, x, obj, _ = lp_solve(c, A, b, equality_constraints)
        if st != 0:
            return None
        return OracleLPOutput(obj, x)


class OracleProblem:
    """A presolved program held in the layout the C oracle expects."""

    def __init__(self, A, b, F, c, H, Q, A_t, b_t, n_eq):
        f = lambda a: numpy.ascontiguousarray(a, dtype=numpy.float64)
        self.A, self.b, self.F, self.c, self.H = f(A), f(b).reshape(-1), f(F), f(c).reshape(-1), f(H)
        self.A_t, self.b_t = f(A_t), f(b_t).reshape(-1)
        self.n_c, self.n_x = self.A.shape
        self.n_t = self.F.shape[1]
        self.n_tc = self.A_t.shape[0]
        self.n_eq = int(n_eq)
        self.is_qp = Q is not None
        self.Q = f(Q) if self.is_qp else numpy.zeros((self.n_x, self.n_x))
        self.cs = _Problem(self.n_x, self.n_t, self.n_c, self.n_eq, self.n_tc, int(self.is_qp), _dp(self.A),
                           _dp(self.b), _dp(self.F), _dp(self.c), _dp(self.H), _dp(self.Q), _dp(self.A_t),
                           _dp(self.b_t))
        self.rec_d = int(lib().orc_region_doubles(ctypes.byref(self.cs)))
        self.rec_i = int(lib().orc_region_ints(ctypes.byref(self.cs)))

    # --- program primitives -------------------------------------------------------------------------------
    def _as(self, active_set):
        return numpy.ascontiguousarray(active_set, dtype=numpy.int32)

    def check_feasibility(self, active_set, check_rank=True) -> bool:
        a = self._as(active_set)
        return lib().orc_check_feasibility(ctypes.byref(self.cs), _ip(a), len(a), int(check_rank)) == 1

    def check_optimality(self, active_set) -> bool:
        a = self._as(active_set)
        return lib().orc_check_optimality(ctypes.byref(self.cs), _ip(a), len(a)) == 1

    def is_full_rank(self, active_set) -> bool:
        a = self._as(active_set)
        return lib().orc_is_full_rank(_dp(self.A), self.n_x, _ip(a), len(a)) == 1

    def optimal_control_law(self, active_set):
        a = self._as(active_set)
        k = len(a)
        A_x, b_x = numpy.zeros((self.n_x, self.n_t)), numpy.zeros(self.n_x)
        A_l, b_l = numpy.zeros((k, self.n_t)), numpy.zeros(k)
        st = lib().orc_optimal_control_law(ctypes.byref(self.cs), _ip(a), k, _dp(A_x), _dp(b_x), _dp(A_l), _dp(b_l))
        if st:
            raise numpy.linalg.LinAlgError('Singular matrix')
        return A_x, b_x.reshape(-1, 1), A_l, b_l.reshape(-1, 1)

    def unpack_region(self, rec_d: numpy.ndarray, rec_i: numpy.ndarray) -> Dict:
        nx, nt, nc, ntc = self.n_x, self.n_t, self.n_c, self.n_tc
        kmax, emax = nc, nc + ntc
        k, nE, n_om, n_la, n_re = (int(v) for v in rec_i[:5])
        o = 0
        A_x = rec_d[o:o + nx * nt].reshape(nx, nt); o += nx * nt
        b_x = rec_d[o:o + nx]; o += nx
        A_l = rec_d[o:o + kmax * nt].reshape(kmax, nt)[:k]; o += kmax * nt
        b_l = rec_d[o:o + kmax][:k]; o += kmax
        E = rec_d[o:o + emax * nt].reshape(emax, nt)[:nE]; o += emax * nt
        f = rec_d[o:o + emax][:nE]
        q = 5
        act = rec_i[q:q + kmax][:k]; q += kmax
        om = rec_i[q:q + ntc][:n_om]; q += ntc
        la = rec_i[q:q + kmax][:n_la]; q += kmax
        ridx = rec_i[q:q + nc][:n_re]; q += nc
        rcon = rec_i[q:q + nc][:n_re]
        return {'A': A_x.copy(), 'b': b_x.copy().reshape(-1, 1), 'C': A_l.copy(), 'd': b_l.copy().reshape(-1, 1),
                'E': E.copy(), 'f': f.copy().reshape(-1, 1), 'active_set': act.tolist(), 'omega_set': om.tolist(),
                'lambda_set': la.tolist(), 'regular_set': [ridx.tolist(), rcon.tolist()]}

    def full_process(self, active_set) -> int:
        a = self._as(active_set)
        return int(lib().orc_full_process(ctypes.byref(self.cs), _ip(a), len(a), None, None))

    def gen_cr_from_active_set(self, active_set):
        a = self._as(active_set)
        d = numpy.zeros(self.rec_d)
        i = numpy.zeros(self.rec_i, dtype=numpy.int32)
        v = lib().orc_gen_cr(ctypes.byref(self.cs), _ip(a), len(a), _dp(d), _ip(i))
        return v, (self.unpack_region(d, i) if v == REGION else None)

    # --- level operator and driver ------------------------------------------------------------------------
    def check_level(self, cands: numpy.ndarray, threads: int = 0, want_regions: bool = True):
        cands = numpy.ascontiguousarray(cands, dtype=numpy.int32)
        n, k = cands.shape
        status = numpy.zeros(n, dtype=numpy.uint8)
        if want_regions:
            d = numpy.zeros((n, self.rec_d))
            i = numpy.zeros((n, self.rec_i), dtype=numpy.int32)
            lib().orc_check_level(ctypes.byref(self.cs), _ip(cands), n, k, threads,
                                  status.ctypes.data_as(_c_uint8_p), _dp(d), _ip(i))
            regions = {j: self.unpack_region(d[j], i[j]) for j in numpy.nonzero(status == REGION)[0]}
            return status, regions
        lib().orc_check_level(ctypes.byref(self.cs), _ip(cands), n, k, threads, status.ctypes.data_as(_c_uint8_p),
                              None, None)
        return status, {}

    def generate_children(self, cands, status, pruned: List[Tuple[int, ...]], mplp_filter=False) -> numpy.ndarray:
        cands = numpy.ascontiguousarray(cands, dtype=numpy.int32)
        n, k = cands.shape
        stride = max([len(p) for p in pruned] + [1])
        pr = numpy.zeros((max(len(pruned), 1), stride), dtype=numpy.int32)
        pk = numpy.zeros(max(len(pruned), 1), dtype=numpy.int32)
        for j, p in enumerate(pruned):
            pr[j, :len(p)] = p
            pk[j] = len(p)
        st = numpy.ascontiguousarray(status, dtype=numpy.uint8)
        args = (ctypes.byref(self.cs), _ip(cands), n, k, st.ctypes.data_as(_c_uint8_p), _ip(pr), _ip(pk), len(pruned),
                stride, int(mplp_filter))
        cnt = lib().orc_generate_children(*args, None, 0)
        out = numpy.zeros((cnt, k + 1), dtype=numpy.int32)
        if cnt:
            lib().orc_generate_children(*args, _ip(out), cnt)
        return out

    def solve(self, threads: int = 0, want_regions: bool = True, max_levels: Optional[int] = None):
        """The parallel combinatorial driver (mpqp_parrallel_combinatorial.py:67-150) on the C primitives.
        Returns (levels, regions, base_verdict); levels = [(cands, status)], regions = list of dicts."""
        e, nc = self.n_eq, self.n_c
        max_depth = max(self.n_x, self.n_t) - e
        base = list(range(e))
        start = base[-1] + 1 if e else 0
        to_check = numpy.array([[*base, i] for i in range(start, nc)], dtype=numpy.int32).reshape(-1, e + 1)
        pruned: List[Tuple[int, ...]] = []
        levels, regions = [], []
        for depth in range(max_depth):
            if max_levels is not None and depth >= max_levels:
                break
            if len(to_check) == 0:
                break
            gen_children = depth + 1 != max_depth
            status, regs = self.check_level(to_check, threads, want_regions)
            levels.append((to_check, status))
            regions.extend(regs[j] for j in sorted(regs))
            if not gen_children:
                break
            kids = self.generate_children(to_check, status, pruned, mplp_filter=not self.is_qp)
            pruned.extend(tuple(int(v) for v in to_check[j])
                          for j in numpy.nonzero((status == INFEASIBLE) | (status == OPTIMAL_NO_REGION)
                                                 | (status == SINGULAR_KKT))[0])
            to_check = kids
        base_as = numpy.array(base, dtype=numpy.int32)
        d = numpy.zeros(self.rec_d)
        i = numpy.zeros(self.rec_i, dtype=numpy.int32)
        bv = lib().orc_full_process(ctypes.byref(self.cs), _ip(base_as), e, _dp(d), _ip(i))
        if bv == REGION:
            regions.append(self.unpack_region(d, i))
        return levels, regions, bv


def problem_from_golden(g) -> OracleProblem:
    """Builds the oracle problem from the *processed* matrices of a golden file."""
    Q = g['raw_Q'] if 'raw_Q' in g.files else None
    return OracleProblem(g['proc_A'], g['proc_b'], g['proc_F'], g['raw_c'], g['raw_H'], Q, g['proc_A_t'],
                         g['proc_b_t'], len(g['proc_eq']))

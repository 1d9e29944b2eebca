"""Import the PPOPT reference (read-only at /root/reference) in THIS container.

TEST INFRASTRUCTURE ONLY.  Nothing under ppopt_amd/ may import this module, and
it cannot run on the GPU box (/root/reference does not exist there).  It is used
by gen_goldens.py to produce the committed fixtures under tests/golden/.

The reference's LP arithmetic lives in third-party native code that is absent
from /root/reference and from this image: GLPK through `cvxopt` (un-pinned:
reference setup.py:22-25, environment.yml:20) or Gurobi (`gurobipy`, un-pinned).
Following SURVEY.md Appendix A we stand in scipy's HiGHS dual simplex for
GLPK at the exact call site the reference uses
(src/ppopt/solver_interface/cvxopt_interface.py:205) and `multiprocess.Pool`
for `pathos.multiprocessing.ProcessingPool`
(src/ppopt/mp_solvers/mpqp_parrallel_combinatorial.py:6).  All Python control
flow, numpy linear algebra, tolerances and bookkeeping stay the reference's own;
no reference file is modified or copied.
"""
import sys
import types

import numpy

REFERENCE_SRC = '/root/reference/src'

LP_CALLS = {'n': 0}


def _install_cvxopt_stub():
    from scipy.optimize import linprog

    mod = types.ModuleType('cvxopt')

    def matrix(a):
        return numpy.asarray(a, dtype=float)

    class _Solvers:
        @staticmethod
        def lp(c, G, h, A=None, b=None, solver=None, options=None):
            # keys consumed by cvxopt_interface.py:18-51
            LP_CALLS['n'] += 1
            c = numpy.asarray(c, float).flatten()
            kw = {}
            if G is not None:
                kw['A_ub'] = numpy.asarray(G, float)
                kw['b_ub'] = numpy.asarray(h, float).flatten()
            if A is not None:
                kw['A_eq'] = numpy.asarray(A, float)
                kw['b_eq'] = numpy.asarray(b, float).flatten()
            res = linprog(c, bounds=(None, None), method='highs-ds', **kw)
            if res.status == 0:
                out = {'status': 'optimal', 'x': res.x, 'primal objective': res.fun}
                out['s'] = res.slack if G is not None else numpy.zeros(0)
                out['z'] = -res.ineqlin.marginals if G is not None else numpy.zeros(0)
                out['y'] = -res.eqlin.marginals if A is not None else numpy.zeros(0)
                return out
            if res.status == 2:
                return {'status': 'primal infeasible'}
            return {'status': 'unknown'}

    mod.matrix = matrix
    mod.solvers = _Solvers
    sys.modules['cvxopt'] = mod


def _install_pathos_stub():
    import multiprocess

    pathos = types.ModuleType('pathos')
    mp = types.ModuleType('pathos.multiprocessing')

    class ProcessingPool:
        def __init__(self, n=None):
            self._n = n

        def map(self, f, xs):
            xs = list(xs)
            if not xs:
                return []
            with multiprocess.Pool(self._n) as pool:
                return pool.map(f, xs)

        def clear(self):
            pass

    mp.ProcessingPool = ProcessingPool
    pathos.multiprocessing = mp
    sys.modules['pathos'] = pathos
    sys.modules['pathos.multiprocessing'] = mp


def load_reference():
    """Returns the imported reference package `ppopt` with the shims in place."""
    sys.dont_write_bytecode = True
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    _install_cvxopt_stub()
    _install_pathos_stub()
    import ppopt  # noqa: F401
    import ppopt.solver as solver_mod
    import ppopt.solver_interface.solver_interface as si
    from ppopt.solver_interface.cvxopt_interface import solve_lp_cvxopt

    # hard-wired 'gurobi' defaults (chebyshev_ball.py:11, critical_region.py:97) -> same LP
    si.solve_lp_gurobi = solve_lp_cvxopt
    solver_mod.solve_lp_gurobi = solve_lp_cvxopt
    return ppopt

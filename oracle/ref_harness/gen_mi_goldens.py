"""Golden vectors for the mixed-integer caller of the path: runs the REAL reference's
`solve_mpmiqp` (mp_solvers/solve_mpmiqp.py:36-66 -> mpmiqp_enumeration.py:12-64 -> mitree.py:22-64) on the
mixed-integer problems its own test-suite holds (tests/test_fixtures.py:176-535, data only) and writes
tests/golden/mi_*.npz.

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).
    python oracle/ref_harness/gen_mi_goldens.py [name ...]

The reference's MILP arithmetic is Gurobi (solver.py:279-280, absent here); scipy's HiGHS branch-and-bound
(`scipy.optimize.milp`) stands in at exactly that call site, the same way HiGHS stands in for GLPK in ref_shims.py.
Everything else (MITree, generate_substituted_problem, presolve, the continuous solve, the 1-D overlap reduction) is
the reference's own code.

What is captured per problem:
  raw_*         constructor inputs (A, b, c, H, [Q], A_t, b_t, F, binary_indices) as the fixture passes them
  proc_*        the program after MPMILP_Program.process_constraints (mpmilp_program.py:73-143)
  combos        feasible binary combinations in the order MITree.get_full_leafs returns them (mitree.py:84-102)
  n_nodes       MITree.count_nodes()
  bin_feas_*    check_bin_feasibility on every partial fixation (mpmilp_program.py:203-237)
  S{i}_*        the substituted continuous program of combination i after its own presolve, and its regions
  P_*           the regions of the enumeration before any overlap reduction, list order
  F_*           the regions of the final Solution (after the 1-D overlap reduction where it applies), list order
  T_*           parameter points with Solution.evaluate / evaluate_objective at them
"""
import os
import sys
import warnings

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import ref_shims  # noqa: E402

ref_shims._install_cvxopt_stub()
ref_shims._install_pathos_stub()
sys.path.insert(0, '/root/reference')

import src.ppopt  # noqa: E402,F401
import src.ppopt.solver as solver_mod  # noqa: E402
import src.ppopt.solver_interface.solver_interface as si  # noqa: E402
from src.ppopt.solver_interface.cvxopt_interface import solve_lp_cvxopt  # noqa: E402
from src.ppopt.solver_interface.solver_interface_utils import SolverOutput  # noqa: E402

si.solve_lp_gurobi = solve_lp_cvxopt
solver_mod.solve_lp_gurobi = solve_lp_cvxopt


def solve_milp_highs(c, A, b, equality_constraints=None, bin_vars=None, verbose=False, get_duals=True):
    """min c'[x,y] s.t. A[x,y] <= b, rows `equality_constraints` as =, y binary, x free (solver.py:248-282)."""
    from scipy.optimize import Bounds, LinearConstraint, milp
    A = numpy.asarray(A, float)
    m, n = A.shape
    bb = numpy.asarray(b, float).flatten()
    cc = numpy.zeros(n) if c is None else numpy.asarray(c, float).flatten()
    lo = numpy.full(m, -numpy.inf)
    eq = list(equality_constraints or [])
    lo[eq] = bb[eq]
    bins = list(bin_vars or [])
    integrality = numpy.zeros(n)
    integrality[bins] = 1
    vlo = numpy.full(n, -numpy.inf)
    vhi = numpy.full(n, numpy.inf)
    vlo[bins] = 0.0
    vhi[bins] = 1.0
    res = milp(cc, constraints=LinearConstraint(A, lo, bb), integrality=integrality, bounds=Bounds(vlo, vhi))
    if res.status != 0 or res.x is None:
        return None
    x = numpy.asarray(res.x, float)
    x[bins] = numpy.round(x[bins])
    slack = bb - A @ x
    return SolverOutput(float(cc @ x), x, slack, numpy.nonzero(numpy.abs(slack) <= 1e-9)[0], None)


si.solve_milp_gurobi = solve_milp_highs
solver_mod.solve_milp_gurobi = solve_milp_highs

import tests.test_fixtures as ref_fixtures  # noqa: E402
from src.ppopt.mp_solvers.mitree import MITree  # noqa: E402
from src.ppopt.mp_solvers.solve_mpmiqp import solve_mpmiqp  # noqa: E402
from src.ppopt.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp  # noqa: E402
from src.ppopt.mpmilp_program import MPMILP_Program  # noqa: E402
from src.ppopt.mpmiqp_program import MPMIQP_Program  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

NAMES = ['simple_mpMILP', 'simple_mpMIQP', 'mpMILP_market_problem', 'mpMIQP_market_problem',
         'bard_mpMILP_adapted', 'bard_mpMILP_adapted_2', 'bard_mpMILP_adapted_degenerate', 'mpMILP_1d',
         'acevedo_mpmilp', 'pappas_multi_objective', 'pappas_multi_objective_2']


# synthetic mixed-integer mpQPs of ppopt_amd.problem_generator.generate_mpmiqp_data (own code): (x, t, m, n_bin, seed)
SYNTHETIC = {'rand_6_3_12_b5_s0': (6, 3, 12, 5, 0), 'rand_4_2_8_b3_s1': (4, 2, 8, 3, 1)}


def synthetic_program(x, t, m, nb, seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location('pg', os.path.join(ROOT, 'ppopt_amd', 'problem_generator.py'))
    pg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pg)
    d = pg.generate_mpmiqp_data(x, t, m, nb, seed)
    return MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])


def capture_ctor_args(fixture_fn):
    """Runs the fixture body with the constructors wrapped so that the raw arguments are recorded."""
    rec = {}
    import inspect

    def wrap(cls):
        orig = cls.__init__
        names = list(inspect.signature(orig).parameters)[1:]

        def init(self, *args, **kw):
            if 'cls' not in rec:
                bound = dict(zip(names, args))
                bound.update(kw)
                rec['cls'] = cls.__name__
                rec['args'] = {k: (None if v is None else numpy.array(v)) for k, v in bound.items()
                               if k not in ('solver', 'post_process')}
            orig(self, *args, **kw)
        cls.__init__ = init
        return orig

    o1 = wrap(MPMIQP_Program)
    o2 = wrap(MPMILP_Program)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            prog = fixture_fn()
    finally:
        MPMIQP_Program.__init__ = o1
        MPMILP_Program.__init__ = o2
    return prog, rec


def pack_regions(prefix, regions, n_x, n_t, with_y=False):
    out = {}
    n = len(regions)
    kmax = max([len(r.active_set) for r in regions], default=0)
    emax = max([r.E.shape[0] for r in regions], default=0)
    out[prefix + 'n'] = numpy.array(n)
    out[prefix + 'k'] = numpy.array([len(r.active_set) for r in regions], dtype=numpy.int32)
    out[prefix + 'as'] = numpy.full((n, kmax), -1, dtype=numpy.int32)
    out[prefix + 'A'] = numpy.zeros((n, n_x, n_t))
    out[prefix + 'b'] = numpy.zeros((n, n_x))
    out[prefix + 'C'] = numpy.zeros((n, kmax, n_t))
    out[prefix + 'd'] = numpy.zeros((n, kmax))
    out[prefix + 'nE'] = numpy.array([r.E.shape[0] for r in regions], dtype=numpy.int32)
    out[prefix + 'E'] = numpy.zeros((n, emax, n_t))
    out[prefix + 'f'] = numpy.zeros((n, emax))
    for i, r in enumerate(regions):
        k = len(r.active_set)
        out[prefix + 'as'][i, :k] = r.active_set
        out[prefix + 'A'][i] = r.A
        out[prefix + 'b'][i] = numpy.asarray(r.b).flatten()
        out[prefix + 'C'][i, :k] = r.C
        out[prefix + 'd'][i, :k] = numpy.asarray(r.d).flatten()
        ne = r.E.shape[0]
        out[prefix + 'E'][i, :ne] = r.E
        out[prefix + 'f'][i, :ne] = numpy.asarray(r.f).flatten()
    if with_y:
        nb = len(regions[0].y_fixation) if n else 0
        out[prefix + 'y'] = numpy.array([list(r.y_fixation) for r in regions], dtype=numpy.int32).reshape(n, nb)
    return out


def theta_box(prog):
    from scipy.optimize import linprog
    nt = prog.num_t()
    lo, hi = numpy.zeros(nt), numpy.zeros(nt)
    for j in range(nt):
        e = numpy.zeros(nt)
        e[j] = 1
        r_lo = linprog(e, A_ub=prog.A_t, b_ub=prog.b_t.flatten(), bounds=(None, None))
        r_hi = linprog(-e, A_ub=prog.A_t, b_ub=prog.b_t.flatten(), bounds=(None, None))
        hi[j] = -r_hi.fun if r_hi.status == 0 else 4.0
        lo[j] = r_lo.fun if r_lo.status == 0 else hi[j] - 8.0      # a parameter space without that bound
    return lo, hi


def generate(name):
    print(f'== {name}', flush=True)
    if name in SYNTHETIC:
        fixture_fn = lambda: synthetic_program(*SYNTHETIC[name])
    else:
        fixture_fn = getattr(ref_fixtures, name)._get_wrapped_function()
    prog, rec = capture_ctor_args(fixture_fn)
    out = {'cls': numpy.array(rec['cls'])}
    for k, v in rec['args'].items():
        if v is not None:
            out['raw_' + k] = v
    for key in ('A', 'b', 'F', 'A_t', 'b_t', 'c', 'H', 'c_c', 'c_t', 'Q_t'):
        out['proc_' + key] = getattr(prog, key)
    if hasattr(prog, 'Q'):
        out['proc_Q'] = prog.Q
    out['proc_eq'] = numpy.array(prog.equality_indices, dtype=numpy.int32)
    out['binary_indices'] = numpy.array(prog.binary_indices, dtype=numpy.int32)
    out['cont_indices'] = numpy.array(prog.cont_indices, dtype=numpy.int32)
    nb = len(prog.binary_indices)

    # partial fixations, every prefix length
    import itertools
    fixes, feas = [], []
    for depth in range(1, nb + 1):  # the reference never tests the empty fixation (mitree.py:49-57)
        for fix in itertools.product([0, 1], repeat=depth):
            fixes.append(list(fix) + [-1] * (nb - depth))
            feas.append(prog.check_bin_feasibility(list(fix)))
    out['bin_fix'] = numpy.array(fixes, dtype=numpy.int32).reshape(len(fixes), nb)
    out['bin_feas'] = numpy.array(feas)

    tree = MITree(prog, depth=0)
    combos = [leaf.fixed_bins for leaf in tree.get_full_leafs()]
    out['combos'] = numpy.array(combos, dtype=numpy.int32).reshape(len(combos), nb)
    out['n_nodes'] = numpy.array(tree.count_nodes())
    print(f'  {rec["cls"]}: n_x {prog.num_x()} n_t {prog.num_t()} n_c {prog.num_constraints()} binaries {nb}, '
          f'{len(combos)} feasible combinations, {tree.count_nodes()} tree nodes', flush=True)

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i, fix in enumerate(combos):
            sub = prog.generate_substituted_problem(fix)
            for key in ('A', 'b', 'F', 'A_t', 'b_t', 'c', 'H', 'c_c', 'c_t'):
                out[f'S{i}_{key}'] = getattr(sub, key)
            if hasattr(sub, 'Q'):
                out[f'S{i}_Q'] = sub.Q
            out[f'S{i}_eq'] = numpy.array(sub.equality_indices, dtype=numpy.int32)
            sol = solve_mpqp(sub, mpqp_algorithm.combinatorial)
            regs = sorted(sol.critical_regions, key=lambda r: tuple(r.active_set))
            out.update(pack_regions(f'S{i}_R_', regs, sub.num_x(), sub.num_t()))
            print(f'    combination {fix}: n_c {sub.num_constraints()} e {len(sub.equality_indices)} -> '
                  f'{len(regs)} regions', flush=True)
        before = solve_mpmiqp(prog, num_cores=1, reduce_overlap=False)
        out.update(pack_regions('P_', before.critical_regions, len(prog.cont_indices), prog.num_t(), with_y=True))
        final = solve_mpmiqp(prog, num_cores=1)
    out['F_overlapping'] = numpy.array(bool(final.is_overlapping))
    out.update(pack_regions('F_', final.critical_regions, len(prog.cont_indices), prog.num_t(), with_y=True))
    print(f'  final: {len(final.critical_regions)} regions, is_overlapping {final.is_overlapping}', flush=True)

    # evaluation samples
    rng = numpy.random.default_rng(11)
    lo, hi = theta_box(prog)
    pts, xs, objs, ok = [], [], [], []
    for _ in range(48):
        th = (lo + (hi - lo) * rng.random(prog.num_t())).reshape(-1, 1)
        x = final.evaluate(th)
        pts.append(th.flatten())
        ok.append(x is not None)
        xs.append(numpy.full(prog.num_x(), numpy.nan) if x is None else numpy.asarray(x).flatten())
        objs.append(numpy.nan if x is None else float(numpy.asarray(final.evaluate_objective(th)).reshape(-1)[0]))
    out['T_theta'] = numpy.array(pts)
    out['T_ok'] = numpy.array(ok)
    out['T_x'] = numpy.array(xs)
    out['T_obj'] = numpy.array(objs)
    print(f'  {sum(ok)}/{len(ok)} sample points inside a region', flush=True)
    numpy.savez_compressed(os.path.join(GOLDEN, 'mi_' + name + '.npz'), **out)


if __name__ == '__main__':
    for nm in (sys.argv[1:] or [*NAMES, *SYNTHETIC]):
        generate(nm)

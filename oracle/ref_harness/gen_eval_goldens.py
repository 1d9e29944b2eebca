"""Evaluation fixtures for CONTINUOUS programs, produced by the REAL reference (imported through ref_shims):
Solution.evaluate / Solution.evaluate_objective (reference src/ppopt/solution.py:60-112) at sample points of the parameter set.

Runs only in the build container (needs /root/reference).  Usage:
    python oracle/ref_harness/gen_eval_goldens.py [name ...]

Per program  tests/golden/eval_<name>.npz:
  raw_*      the constructor inputs (ppopt_amd.problem_generator -- own code), so that the test builds the same program
  T_theta    [m, n_theta] sample points: the Chebyshev centre of every region, that centre moved by 0.9 and by 3 radii in a random
             direction, and uniform points in the (widened) box spanned by the centres -- inside regions, near facets, outside
  T_ok       the reference's Solution.get_region(theta) is not None
  T_x        Solution.evaluate(theta)            (NaN where T_ok is False)
  T_obj      Solution.evaluate_objective(theta)  (NaN where T_ok is False)
  T_region   active set of the region the reference located the point in (padded with -1)
The reference's solve is its own combinatorial algorithm (solve_mpqp(prog, mpqp_algorithm.combinatorial)).
VERDICT r4 item 8(a): until round 5 only the mixed-integer goldens carried such points, and the device's evaluate_batch was compared
with this package's own get_region loop.
"""
import os
import sys
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

from ppopt.mp_solvers.solve_mpqp import mpqp_algorithm, solve_mpqp  # noqa: E402
from ppopt.mplp_program import MPLP_Program  # noqa: E402
from ppopt.mpqp_program import MPQP_Program  # noqa: E402
from ppopt.utils.chebyshev_ball import chebyshev_ball  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

PROBLEMS = {
    'transport_mpqp': lambda: pg.transport_mpqp_data(),
    'rand_5_3_8_s3': lambda: pg.generate_mpqp_data(5, 3, 8, 3),
    'c2_dblint_n5': lambda: pg.double_integrator_data(5),
    'c1_transport_mplp': lambda: pg.transport_mplp_data(),
}


def build(d):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if d['Q'] is None:
            return MPLP_Program(d['A'], d['b'], d['c'], d['H'], d['A_t'], d['b_t'], d['F'], equality_indices=list(d['equality_indices']))
        return MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], equality_indices=list(d['equality_indices']))


def generate(name, n_uniform=96):
    d = PROBLEMS[name]()
    prog = build(d)
    sol = solve_mpqp(prog, mpqp_algorithm.combinatorial)
    regs = sol.critical_regions
    nt = prog.num_t()
    rng = numpy.random.default_rng(23)
    pts, centres = [], []
    for cr in regs:
        cb = chebyshev_ball(cr.E, cr.f, deterministic_solver=prog.solver.solvers['lp'])
        if cb is None:
            continue
        centre, radius = cb.sol[:nt], float(cb.sol[nt])
        centres.append(centre.copy())
        pts.append(centre.copy())
        step = rng.standard_normal(nt)
        pts.append(centre + 0.9 * radius * step / max(numpy.linalg.norm(step), 1e-12))    # still inside
        step = rng.standard_normal(nt)
        pts.append(centre + 3.0 * radius * step / max(numpy.linalg.norm(step), 1e-12))    # probably in a neighbour
    # uniform points in the box spanned by the regions' centres, widened by a quarter on every side (some fall outside every region)
    C = numpy.array(centres)
    lo, hi = C.min(axis=0), C.max(axis=0)
    span = numpy.maximum(hi - lo, 1.0)
    lo, hi = lo - 0.25 * span, hi + 0.25 * span
    pts += [lo + (hi - lo) * rng.random(nt) for _ in range(n_uniform)]
    ok, xs, objs, act = [], [], [], []
    kmax = max([len(r.active_set) for r in regs] + [1])
    for th in pts:
        t = numpy.asarray(th, dtype=float).reshape(-1, 1)
        cr = sol.get_region(t)
        ok.append(cr is not None)
        if cr is None:
            xs.append(numpy.full(prog.num_x(), numpy.nan))
            objs.append(numpy.nan)
            act.append([-1] * kmax)
        else:
            xs.append(numpy.asarray(sol.evaluate(t)).flatten())
            objs.append(float(numpy.asarray(sol.evaluate_objective(t)).reshape(-1)[0]))
            act.append(list(cr.active_set) + [-1] * (kmax - len(cr.active_set)))
    out = {f'raw_{k}': (numpy.array([]) if v is None else numpy.asarray(v)) for k, v in d.items() if k != 'equality_indices'}
    out['raw_has_Q'] = numpy.array(d['Q'] is not None)
    out['raw_equality_indices'] = numpy.array(list(d['equality_indices']), dtype=numpy.int32)
    out['T_theta'] = numpy.array(pts)
    out['T_ok'] = numpy.array(ok)
    out['T_x'] = numpy.array(xs)
    out['T_obj'] = numpy.array(objs)
    out['T_region'] = numpy.array(act, dtype=numpy.int32)
    out['n_regions'] = numpy.array(len(regs))
    numpy.savez_compressed(os.path.join(GOLDEN, 'eval_' + name + '.npz'), **out)
    print(f'{name}: {len(regs)} regions, {len(pts)} points, {sum(ok)} inside a region', flush=True)


if __name__ == '__main__':
    for nm in (sys.argv[1:] or list(PROBLEMS)):
        generate(nm)

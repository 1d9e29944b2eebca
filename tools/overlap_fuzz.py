"""Random small mpQPs solved three ways -- default (region stage under the (x,theta) stage where it pays), MPC_NO_ROVERLAP=1, and
MPC_TEST_LATE=3 with MPC_ROVERLAP_MIN=0 (every level overlaps and leaves three optimal candidates to the late path) -- must give the
same regions (run on the GPU box):  python tools/overlap_fuzz.py [n_programs] [seed]"""
import os, sys, warnings
sys.path.insert(0, '.')
import numpy
from ppopt_amd.problem_generator import generate_mpqp
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial

n_prog = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = numpy.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
MODES = [{}, {'MPC_NO_ROVERLAP': '1'}, {'MPC_TEST_LATE': '3', 'MPC_ROVERLAP_MIN': '0'}, {'MPC_TEST_LATE': '1000000', 'MPC_ROVERLAP_MIN': '0'}]
bad = 0
edge = 0
edge_slacks = []
total_regions = 0


def knife_edge(r1, r2):
    """(True, slacks) when the two records of one region agree in the laws and differ only in rows that do not cut the region:
    for every row that one record has and the other lacks, the minimum of its slack over the region described by the SHORTER
    facet list is reported (>= -1e-6: the row is redundant there -- weakly when ~0, the reference keeps such rows, strongly when
    clearly positive, the reference drops them; which side of the 1e-7 LP tolerance a facet test lands on depends on the engine
    and the walk order).  A clearly negative value would mean a lost facet."""
    from scipy.optimize import linprog
    for f in ('A', 'b', 'C', 'd'):
        if getattr(r1, f).shape != getattr(r2, f).shape or not numpy.allclose(getattr(r1, f), getattr(r2, f), rtol=0, atol=1e-8):
            return False, []
    rows1 = [numpy.append(e, v) for e, v in zip(r1.E, r1.f.ravel())]
    rows2 = [numpy.append(e, v) for e, v in zip(r2.E, r2.f.ravel())]

    def missing(rows_a, rows_b):
        return [ra for ra in rows_a if not any(numpy.abs(ra - rb).max() < 1e-7 for rb in rows_b)]
    small = r1 if len(rows1) <= len(rows2) else r2
    slacks = []
    for row in missing(rows1, rows2) + missing(rows2, rows1):
        e, v = row[:-1], row[-1]
        res = linprog(-e, A_ub=small.E, b_ub=small.f.ravel(), bounds=(None, None), method='highs')   # max e.theta => min slack
        slacks.append(float('nan') if res.status != 0 else v + res.fun)
    if all(sl == sl and sl >= -1e-6 for sl in slacks):
        return True, slacks
    # a row that cuts: only on a sliver -- a region whose inscribed radius is within two orders of the LP tolerance (1e-7), where
    # "feasible within the tolerance" and "feasible" describe different sets (the facet tests of the reference itself, run with
    # GLPK / HiGHS tolerances, are decided by the tolerance there)
    big = r1 if len(rows1) >= len(rows2) else r2
    nrm = numpy.linalg.norm(big.E, axis=1)
    c = numpy.zeros(big.E.shape[1] + 1); c[-1] = -1.0
    res = linprog(c, A_ub=numpy.hstack([big.E, nrm.reshape(-1, 1)]), b_ub=big.f.ravel(), bounds=[(None, None)] * big.E.shape[1] + [(0, None)], method='highs')
    radius = res.x[-1] if res.status == 0 else float('nan')
    slacks.append(('chebyshev radius', radius))
    return bool(radius == radius and radius < 1e-5), slacks


for p in range(n_prog):
    nx, nt = int(rng.integers(3, 9)), int(rng.integers(2, 6))
    m = int(rng.integers(nx + 2, 3 * nx))
    seed = int(rng.integers(0, 10 ** 6))
    sols = []
    for env in MODES:
        for key in ('MPC_NO_ROVERLAP', 'MPC_TEST_LATE', 'MPC_ROVERLAP_MIN'):
            os.environ.pop(key, None)
        os.environ.update(env)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            prog = generate_mpqp(nx, nt, m, seed)
        sol = mpqp_hip_combinatorial.solve(prog)
        sols.append({tuple(r.active_set): r for r in sol.critical_regions})
        prog.release_engine()
    base = sols[0]
    total_regions += len(base)
    for env, other in zip(MODES[1:], sols[1:]):
        if set(other) != set(base):
            bad += 1
            print('REGION SETS DIFFER', (nx, nt, m, seed), env, len(base), len(other), sorted(set(base) ^ set(other))[:5])
            continue
        for key, r1 in base.items():
            r2 = other[key]
            ok = r1.omega_set == r2.omega_set and r1.lambda_set == r2.lambda_set and r1.regular_set == r2.regular_set
            ok = ok and all(numpy.allclose(getattr(r1, f), getattr(r2, f), rtol=0, atol=1e-8) for f in ('A', 'b', 'C', 'd', 'E', 'f'))
            ke, slacks = (False, []) if ok else knife_edge(r1, r2)
            if not ok and ke:
                edge += 1
                edge_slacks.extend(slacks)
                continue
            if not ok:
                print('   slacks of the rows in dispute:', slacks)
                bad += 1
                what = [f for f in ('omega_set', 'lambda_set', 'regular_set') if getattr(r1, f) != getattr(r2, f)]
                for f in ('A', 'b', 'C', 'd', 'E', 'f'):
                    a1, a2 = getattr(r1, f), getattr(r2, f)
                    if a1.shape != a2.shape:
                        what.append(f'{f} shape {a1.shape} vs {a2.shape}')
                    elif not numpy.allclose(a1, a2, rtol=0, atol=1e-8):
                        what.append(f'{f} max diff {numpy.abs(a1 - a2).max():.2e}')
                print('REGION DIFFERS', (nx, nt, m, seed), env, key, what, 'omega', r1.omega_set, r2.omega_set, 'regular', r1.regular_set, r2.regular_set)
                break
print('minimum slacks of the disputed rows:', [v if isinstance(v, tuple) else round(v, 9) for v in edge_slacks])
print(f'{n_prog} programs x {len(MODES)} modes, {total_regions} regions, {edge} regions whose facet lists differ in rows that are redundant within tolerance or on slivers, {bad} other differences')

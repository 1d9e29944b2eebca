"""Medium-size random mpQPs (levels of 1e4-1e6 candidates, where the region stage really runs under the (x,theta) stage and the base
set on the twin handle): default against MPC_NO_ROVERLAP=1 + no twin -- same region sets, same laws (run on the GPU box).
python tools/overlap_big.py [n_programs] [levels]"""
import os, sys, time, warnings
sys.path.insert(0, '.')
import numpy
from ppopt_amd.problem_generator import generate_mpqp
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
n_prog = int(sys.argv[1]) if len(sys.argv) > 1 else 6
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = numpy.random.default_rng(3)
bad = 0
for p in range(n_prog):
    nx, nt = int(rng.integers(12, 21)), int(rng.integers(4, 9))
    m = int(rng.integers(12, 21))
    seed = int(rng.integers(0, 10 ** 6))
    out = []
    for env, twin in (({}, True), ({'MPC_NO_ROVERLAP': '1'}, False)):
        os.environ.pop('MPC_NO_ROVERLAP', None)
        os.environ.update(env)
        mpqp_hip_combinatorial.BASE_ON_TWIN = twin
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            prog = generate_mpqp(nx, nt, m, seed)
        prof = []
        mpqp_hip_combinatorial.solve(prog, max_levels=levels)
        t = time.perf_counter()
        sol = mpqp_hip_combinatorial.solve(prog, max_levels=levels, profile=prof)
        dt = time.perf_counter() - t
        out.append(({tuple(r.active_set): r for r in sol.critical_regions}, dt, sum(q['candidates'] for q in prof)))
        prog.release_engine()
    (a, ta, na), (b, tb, nb) = out
    same = set(a) == set(b) and na == nb
    laws = same and all(numpy.allclose(a[k].A, b[k].A, atol=1e-8, rtol=0) and numpy.allclose(a[k].b, b[k].b, atol=1e-8, rtol=0) for k in a)
    rows = sum(1 for k in a if k in b and a[k].E.shape != b[k].E.shape)
    bad += 0 if (same and laws) else 1
    print(f'({nx},{nt},{m},{seed}) n_c={prog.num_constraints()} candidates {na} regions {len(a)} | overlap {ta*1e3:.2f} ms, sequential {tb*1e3:.2f} ms | sets equal {same}, laws equal {laws}, facet lists of different length {rows}')
print('programs with different region sets or laws:', bad)

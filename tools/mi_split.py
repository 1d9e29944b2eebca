"""Where the time of the mixed-integer enumeration goes (bench's `mi_enumeration` extra: generate_mpmiqp_data(8,4,16,6,1)):
one sub-program after the other on one thread, wall time per stage summed over the sub-programs, level statistics.
usage: python tools/mi_split.py [x t m n_bin seed]"""
import os
import sys
import time
import warnings
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ppopt_amd import MPMIQP_Program  # noqa: E402
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial  # noqa: E402
from ppopt_amd.problem_generator import generate_mpmiqp_data  # noqa: E402

args = [int(v) for v in sys.argv[1:]]
x, t, m, nb, seed = (args + [8, 4, 16, 6, 1][len(args):])[:5]
d = generate_mpmiqp_data(x, t, m, nb, seed)
warnings.simplefilter('ignore')
prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
combos = prog.feasible_combinations()
for rep in range(2):
    tot = defaultdict(float)
    lev_n, lev_ms, lev_cnt, n_reg = defaultdict(int), defaultdict(float), defaultdict(int), 0
    t_all = time.perf_counter()
    for fix in combos:
        t0 = time.perf_counter()
        sub = prog.generate_substituted_problem(fix)
        t1 = time.perf_counter()
        eng = sub.engine(0)
        t2 = time.perf_counter()
        prof = []
        sol = mpqp_hip_combinatorial.solve(sub, profile=prof, prune_lowdim=False)
        t3 = time.perf_counter()
        for cr in sol.critical_regions:
            cr.materialize() if hasattr(cr, 'materialize') else None
        t4 = time.perf_counter()
        sub.release_engine()
        t5 = time.perf_counter()
        tot['substitute+presolve'] += t1 - t0; tot['engine create'] += t2 - t1; tot['solve'] += t3 - t2
        tot['materialise regions'] += t4 - t3; tot['destroy'] += t5 - t4
        n_reg += len(sol.critical_regions)
        for p in prof:
            if p['depth'] > 0:
                lev_n[p['depth']] += p['candidates']; lev_ms[p['depth']] += p.get('ms_wall', 0.0); lev_cnt[p['depth']] += 1
                tot['kernel ms (events)'] += 1e-3 * (p.get('ms_verdict', 0) + p.get('ms_region', 0) + p.get('ms_children', 0))
                tot['(x,theta) runs flagged doubtful (count/1000)'] += 1e-3 * p.get('xtheta_fallbacks', 0) * 1e-3
                tot['candidates (count/1000)'] += 1e-3 * p['candidates'] * 1e-3
    wall = time.perf_counter() - t_all
    print(f'rep {rep}: {len(combos)} sub-programs, {n_reg} regions, {1e3 * wall:.1f} ms sequential; per stage (ms): '
          + ', '.join(f'{k} {1e3 * v:.1f}' for k, v in tot.items()))
    print('   levels: ' + ', '.join(f'L{dep}: {lev_cnt[dep]} runs, {lev_n[dep] // max(lev_cnt[dep], 1)} cand avg, {lev_ms[dep] / max(lev_cnt[dep], 1):.3f} ms avg' for dep in sorted(lev_n)))

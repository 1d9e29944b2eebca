"""Wall time of the REAL reference's solve_mpmiqp on the synthetic mixed-integer mpQP that tools/mi_run.py times on
the MI355X (build container only; LP/MILP arithmetic through the HiGHS stand-ins of gen_mi_goldens.py).
    python oracle/ref_harness/time_mi_reference.py [x t m n_bin seed]"""
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_mi_goldens as G  # noqa: E402  (installs the shims, imports src.ppopt)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location('pg', os.path.join(G.ROOT, 'ppopt_amd', 'problem_generator.py'))
pg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(pg)

args = [int(v) for v in sys.argv[1:]]
x, t, m, nb, seed = (args + [6, 3, 12, 5, 0][len(args):])[:5]
d = pg.generate_mpmiqp_data(x, t, m, nb, seed)
warnings.simplefilter('ignore')
t0 = time.perf_counter()
prog = G.MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
t1 = time.perf_counter()
sol = G.solve_mpmiqp(prog, num_cores=1)
t2 = time.perf_counter()
print(f'reference: presolve {t1 - t0:.2f} s, solve_mpmiqp {t2 - t1:.2f} s, {len(sol.critical_regions)} regions '
      f'({len(set(tuple(r.y_fixation) for r in sol.critical_regions))} fixations with regions)')

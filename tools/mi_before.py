"""What the 64 constructions of the mixed-integer bench workload cost before the shared solve: sequential (one thread: substitution +
presolve, then set-up) against the threaded form with coalesced presolve LPs.  python tools/mi_before.py"""
import sys, time, warnings, threading
sys.path.insert(0, '.')
from ppopt_amd import MPMIQP_Program
from ppopt_amd.problem_generator import generate_mpmiqp_data
from ppopt_amd.solver import LPCoalescer
import copy
d = generate_mpmiqp_data(8, 4, 16, 6, 1)
warnings.simplefilter('ignore')
prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
fixes = prog.feasible_combinations()
for rep in range(3):
    t0 = time.perf_counter()
    subs = [prog.generate_substituted_problem(f) for f in fixes]
    t1 = time.perf_counter()
    for s in subs: s.engine(0, closed=True)
    t2 = time.perf_counter()
    for s in subs: s.release_engine()
    # threaded + coalescer, timing the two parts inside the threads
    co = LPCoalescer(prog.solver, len(fixes)); parked = copy.copy(prog); parked.solver = co.solver()
    tsub = [0.0] * len(fixes); teng = [0.0] * len(fixes); res = [None] * len(fixes)
    def run(j):
        a = time.perf_counter()
        try:
            sub = parked.generate_substituted_problem(fixes[j])
        finally:
            co.worker_done()
        b = time.perf_counter()
        sub.solver = prog.solver
        sub.engine(0, closed=True)
        c = time.perf_counter()
        tsub[j], teng[j], res[j] = b - a, c - b, sub
    t3 = time.perf_counter()
    th = [threading.Thread(target=run, args=(j,)) for j in range(len(fixes))]
    for x in th: x.start()
    for x in th: x.join()
    t4 = time.perf_counter()
    for s in res: s.release_engine()
    print('sequential: substitute+presolve %.1f ms, set-up %.1f ms | threaded: wall %.1f ms (per thread: construction %.1f..%.1f ms, set-up %.2f..%.2f ms; LP flushes %d for %d calls)'
          % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t4 - t3), 1e3 * min(tsub), 1e3 * max(tsub), 1e3 * min(teng), 1e3 * max(teng), co.n_flushes, co.n_calls))

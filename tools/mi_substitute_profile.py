import sys, time, warnings, cProfile, pstats
sys.path.insert(0, '/root/repo')
import os
os.chdir(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, '.')
from ppopt_amd import MPMIQP_Program
from ppopt_amd.problem_generator import generate_mpmiqp_data
d = generate_mpmiqp_data(8, 4, 16, 6, 1)
warnings.simplefilter('ignore')
prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
fixes = prog.feasible_combinations()
def substitute(fixes):
    subs = [prog.generate_substituted_problem(fix, deferred=True) for fix in fixes]
    t1 = time.perf_counter()
    requests = [sub._redundancy_request() for sub in subs]
    t2 = time.perf_counter()
    answers = prog.solver.lp_feasible_many([(PA, Pb, [[*eq, i] for i in todo]) for PA, Pb, eq, todo in requests])
    t3 = time.perf_counter()
    for sub, req, ok in zip(subs, requests, answers):
        sub._redundancy_apply(req, ok.tolist())
    t4 = time.perf_counter()
    for sub in subs:
        sub.engine(0, closed=True)
    t5 = time.perf_counter()
    return subs, (t1, t2, t3, t4, t5)
for rep in range(4):
    t0 = time.perf_counter()
    subs, (t1, t2, t3, t4, t5) = substitute(fixes)
    print('substitute %.1f ms: generate %.1f, requests %.1f, LP batch %.1f, apply %.1f, engines (one thread) %.1f' % tuple(1e3 * v for v in (t5 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
    for s in subs: s.release_engine()
pr = cProfile.Profile(); pr.enable(); subs, _ = substitute(fixes); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)

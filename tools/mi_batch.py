"""Mixed-integer enumeration, bench workload (generate_mpmiqp_data(8,4,16,n_bin=6,seed=1): 64 fixations): sub-programs solved together
(mpc_level_run_batch) against one by one; stage split of the batched form.  usage: python tools/mi_batch.py [x t m n_bin seed]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ppopt_amd import MPMIQP_Program  # noqa: E402
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial  # noqa: E402
from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp  # noqa: E402
from ppopt_amd.problem_generator import generate_mpmiqp_data  # noqa: E402

args = [int(v) for v in sys.argv[1:]]
x, t, m, nb, seed = (args + [8, 4, 16, 6, 1][len(args):])[:5]
d = generate_mpmiqp_data(x, t, m, nb, seed)
warnings.simplefilter('ignore')
prog = MPMIQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], d['binary_indices'])
combos = prog.feasible_combinations()
for rep in range(3):
    t0 = time.perf_counter()
    subs = [prog.generate_substituted_problem(f) for f in combos]
    t1 = time.perf_counter()
    for s in subs:
        s.engine(0)
    t2 = time.perf_counter()
    prof = []
    sols = mpqp_hip_combinatorial.solve_many(subs, profile=prof)
    t3 = time.perf_counter()
    for s in subs:
        s.release_engine()
    t4 = time.perf_counter()
    print(f'rep {rep}: {len(subs)} sub-programs, {sum(len(s) for s in sols)} regions; substitute+presolve {1e3*(t1-t0):.1f} ms, set-up {1e3*(t2-t1):.1f}, '
          f'solve_many {1e3*(t3-t2):.1f}, release {1e3*(t4-t3):.1f}')
    print(f'   device, shared levels: {sum(p.get("ms_launches", 0) for p in prof):.1f} ms')
    print('   levels: ' + ', '.join(f"L{p['depth']}: {p['members']}/{p['shared_launches']} members, {p['candidates']} cand, {p['regions']} reg, "
                                    f"launches {p.get('ms_launches', 0):.2f} ms, wait {p.get('ms_wait', 0):.2f}, wall {p['ms_wall']:.2f} ms" for p in prof))
from ppopt_amd.mp_solvers import mpmiqp_enumeration  # noqa: E402
for env in (() if os.environ.get('MI_BATCH_ONLY') else ('0', '1', 'L', '0', 'L', '0', '1')):
    os.environ['MPC_NO_LP_COALESCE'] = '1' if env == 'L' else '0'
    env = '0' if env == 'L' else env
    if env[0] == 'c':
        mpmiqp_enumeration.BATCH_CHUNKS = int(env[1:]); env = '0'
    os.environ['MPC_NO_BATCH'] = env
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        sol = solve_mpmiqp(prog)
        best = min(best, time.perf_counter() - t0)
    print(f'solve_mpmiqp MPC_NO_BATCH={env} MPC_NO_LP_COALESCE={os.environ["MPC_NO_LP_COALESCE"]}: {1e3*best:.1f} ms, {len(sol)} regions, {len(combos)/best:.0f} sub-programs/s')

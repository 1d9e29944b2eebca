"""The region record of ONE active set under the A/B switches of the region kernel (run on the GPU box):
python tools/one_region.py nx nt m seed  i j k ..."""
import os, sys, warnings
sys.path.insert(0, '.')
import numpy
from ppopt_amd.problem_generator import generate_mpqp
nx, nt, m, seed = (int(v) for v in sys.argv[1:5])
aset = [int(v) for v in sys.argv[5:]]
numpy.set_printoptions(precision=6, linewidth=220)
for env in ({}, {'MPC_NO_RBOX': '1'}, {'MPC_NO_RSPLIT': '1'}, {'MPC_NO_RBOX': '1', 'MPC_NO_RSPLIT': '1'}, {'MPC_FORCE_V1': '1'}):
    for key in ('MPC_NO_RBOX', 'MPC_NO_RSPLIT', 'MPC_FORCE_V1'):
        os.environ.pop(key, None)
    os.environ.update(env)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = generate_mpqp(nx, nt, m, seed)
    eng = prog.engine(0)
    from ppopt_amd.mp_solvers.mpqp_hip_combinatorial import unpack_regions
    cand = numpy.array([aset], dtype=numpy.int32)
    status, rd, ri, _, _ = eng.check_level(cand, numpy.zeros((0, eng.mask_words), dtype=numpy.uint64), False)
    regs = unpack_regions(rd, ri, eng.n_x, eng.n_t, eng.n_c, eng.n_tc) if len(rd) else []
    print(env, 'status', status.tolist(), 'rows', regs[0].E.shape[0] if regs else None, 'lambda', regs[0].lambda_set if regs else None, 'regular', regs[0].regular_set if regs else None)
    if regs and not env:
        print(numpy.hstack([regs[0].E, regs[0].f]))
    prog.release_engine()

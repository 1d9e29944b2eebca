import sys, warnings
sys.path.insert(0, '.')
import numpy
from ppopt_amd import MPQP_Program, problem_generator as pg
d = pg.generate_mpqp_data(7, 4, 24, 623692)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
eng = prog.engine()
for then_base in (False, True):
    eng.pruned_clear(); eng.frontier_root()
    for lv in range(7):
        gen = lv != 6
        n, k = eng.frontier_info()
        eng.level_start(gen, stream=True, then_base=then_base and not gen)
        info = eng.level_stream_info()
        got = None
        if info is not None:
            hd, hi, er, chunk, n_chunks = info
            for j in range(n_chunks):
                eng.level_chunk_wait(j)
            got = (len(hi), chunk, n_chunks, int((hi[:, 0] == 3).sum()), numpy.bincount(numpy.clip(hi[:, 0], 0, 9), minlength=10).tolist())
        st = eng.level_wait()
        print('then_base', then_base, 'level', lv + 1, 'n', n, 'regions', st.n_regions, 'n_opt', st.n_opt, 'retry', st.n_region_retry, 'stream', got)
        if gen: eng.frontier_advance()

import sys, time, numpy
sys.path.insert(0, '.')
from ppopt_amd import _lib
from oracle import oracle as orc

def run(name, max_levels=None):
    g = numpy.load(f'tests/golden/{name}.npz')
    Q = g['raw_Q'] if 'raw_Q' in g.files else None
    eng = _lib.Engine(g['proc_A'], g['proc_b'], g['proc_F'], g['raw_c'], g['raw_H'], Q, g['proc_A_t'], g['proc_b_t'], len(g['proc_eq']))
    nl = int(g['n_levels'])
    max_depth = max(eng.n_x, eng.n_t) - eng.n_eq
    eng.frontier_root()
    tot = 0; mism = 0; t0 = time.time(); ms = 0.0; piv = 0; nreg = 0
    extra = int(sys.argv[sys.argv.index('--extra') + 1]) if '--extra' in sys.argv else 1
    if not bool(g['complete']): max_depth = min(max_depth, nl + extra)
    for depth in range(max_depth):
        gen = depth + 1 != max_depth
        st = eng.level_run(gen)
        tot += st.n; ms += st.ms_total; piv += st.lp_pivots; nreg += st.n_regions
        status = eng.level_status()
        cands = eng.frontier_get()
        if depth < nl:
            gc, gv = g[f'L{depth}_cands'], g[f'L{depth}_verdict']
            if cands.shape != gc.shape or not numpy.array_equal(cands, gc):
                print(name, 'level', depth, 'CANDIDATE LISTS DIFFER', cands.shape, gc.shape)
            else:
                d = numpy.nonzero(status != gv)[0]
                mism += len(d)
                for j in d[:4]:
                    print(name, 'level', depth, 'cand', cands[j], 'gpu', status[j], 'ref', gv[j])
        print(f'  L{depth}: n={st.n} k={st.k} hist={list(st.n_status)} children={st.n_children} pruned_new={st.n_pruned_new} pivots={st.lp_pivots} xlp={st.n_xtheta_lp}/{st.n_xtheta_fallback}/c{st.n_x_cached} rretry={st.n_region_retry} cyc(kkt/th/x)={[int(v)//max(int(st.n),1) for v in st.wave_cycles[:3]]} ms v/r/c = {st.ms_verdict:.3f}/{st.ms_region:.3f}/{st.ms_children:.3f}')
        if not gen or st.n_children == 0: break
        eng.frontier_advance()
    wall = time.time() - t0
    print(f'{name}: {tot} candidates, {nreg} regions (ref {len(g["R_k"])} incl base), mismatches {mism}, kernel ms {ms:.2f}, wall {wall*1e3:.1f} ms, {tot/max(ms,1e-9)*1e3:.0f} cand/s (kernel) pivots/cand {piv/max(tot,1):.1f}, lds {eng.lds_bytes(0)}/{eng.lds_bytes(1)}')
    eng.close()

print(_lib.load().mpc_version(), 'devices', _lib.load().mpc_device_count())
# generic LPs vs goldens
g = numpy.load('tests/golden/lp_cases.npz')
bad = 0
for i in range(int(g['n'])):
    A, b, c, eq = g[f'lp{i}_A'], g[f'lp{i}_b'], g[f'lp{i}_c'], g[f'lp{i}_eq']
    fl = numpy.zeros((1, A.shape[0]), dtype=numpy.uint8); fl[0, eq] = 1
    st, x, obj, it = _lib.lp_solve_batch(A[None], b.reshape(1, -1), c.reshape(1, -1), fl)
    ost, ox, oobj, oit = orc.lp_solve(c, A, b, eq)
    ok = bool(g[f'lp{i}_ok'])
    if (st[0] == 0) != ok or st[0] != ost or it[0] != oit or (ok and obj[0] != oobj):
        bad += 1
        if bad < 6: print('LP', i, 'gpu', st[0], obj[0], it[0], 'oracle', ost, oobj, oit, 'ref ok', ok)
print('lp_cases mismatches (status/iters/obj bitwise vs oracle):', bad)
for name in sys.argv[1:]:
    if name == '--extra': break
    run(name)

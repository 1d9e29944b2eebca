"""A/B of the register-engine kernels (default) against the LDS-engine kernels (MPC_FORCE_V1=1) per candidate."""
import os, sys, subprocess, numpy
sys.path.insert(0, '.')
name = sys.argv[1]; levels = int(sys.argv[2])
if len(sys.argv) > 3:
    from ppopt_amd import _lib
    g = numpy.load(f'tests/golden/{name}.npz')
    Q = g['raw_Q'] if 'raw_Q' in g.files else None
    eng = _lib.Engine(g['proc_A'], g['proc_b'], g['proc_F'], g['raw_c'], g['raw_H'], Q, g['proc_A_t'], g['proc_b_t'], len(g['proc_eq']))
    eng.frontier_root()
    out = {}
    for d in range(levels):
        st = eng.level_run(True)
        out[f'c{d}'] = eng.frontier_get(); out[f's{d}'] = eng.level_status()
        eng.frontier_advance()
    numpy.savez(sys.argv[3], **out)
    sys.exit(0)
env = dict(os.environ); env['MPC_FORCE_V1'] = '1'
subprocess.check_call([sys.executable, __file__, name, str(levels), '/tmp/ab_v1.npz'], env=env)
env['MPC_FORCE_V1'] = '0'
subprocess.check_call([sys.executable, __file__, name, str(levels), '/tmp/ab_v2.npz'], env=env)
a, b = numpy.load('/tmp/ab_v1.npz'), numpy.load('/tmp/ab_v2.npz')
for d in range(levels):
    if a[f'c{d}'].shape != b[f'c{d}'].shape or not numpy.array_equal(a[f'c{d}'], b[f'c{d}']):
        print('level', d, 'frontiers differ'); break
    diff = numpy.nonzero(a[f's{d}'] != b[f's{d}'])[0]
    print('level', d, 'n', len(a[f's{d}']), 'diffs', len(diff), 'v1 hist', numpy.bincount(a[f's{d}'], minlength=7), 'v2 hist', numpy.bincount(b[f's{d}'], minlength=7))
    for j in diff[:6]:
        print('   cand', a[f'c{d}'][j], 'v1', a[f's{d}'][j], 'v2', b[f's{d}'][j])

"""Config 5: the active sets on which the device reports MPC_SINGULAR_KKT (4) where the reference says "feasible, not optimal" (1), among
the candidates of tests/golden/c5_deep.npz the reference decided before a KKT solve or with cond(KKT) < 1e10.  Writes the explicit list the
deep test accepts (tests/golden/c5_singular_for_not_optimal.json).  Run on the GPU box: python tools/c5_singular_list.py"""
import json, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy
from conftest import load_golden
from test_gpu_parity import engine_from_golden, run_levels
g = load_golden('c5_control_allocation'); d = load_golden('c5_deep')
eng = engine_from_golden(g)
levels, _ = run_levels(eng)
got = {tuple(c): int(v) for cands, status, _ in levels for c, v in zip(cands.tolist(), status.tolist())}
out = []
for i in range(int(d['n_levels'])):
    for cand, v, cond in zip(d[f'L{i}_cands'].tolist(), d[f'L{i}_verdict'].tolist(), d[f'L{i}_cond'].tolist()):
        key = tuple(cand)
        if (numpy.isnan(cond) or cond < 1e10) and key in got and got[key] == 4 and int(v) == 1:
            out.append(list(key))
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'note': 'config 5: device MPC_SINGULAR_KKT (4) where the reference says 1 (feasible, not optimal); tools/c5_singular_list.py', 'active_sets': sorted(out)},
          open('gpurun_out/c5_singular_for_not_optimal.json', 'w'))
print(len(out), 'active sets')

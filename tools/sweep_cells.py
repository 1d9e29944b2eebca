"""A few cells of tools/shape_sweep.py under the current environment (A/B of library switches): python tools/sweep_cells.py nx,nt,m [nx,nt,m ...]"""
import json, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import shape_sweep
from ppopt_amd import problem_generator as pg
for arg in sys.argv[1:]:
    nx, nt, mm = (int(v) for v in arg.split(','))
    r = shape_sweep.cell(f'mpqp_{nx}_{nt}_{mm}', pg.generate_mpqp_data(nx, nt, mm, 7))
    print('%-18s cands %8d reg %7d ms %8.2f  %9.3g c/s %6.1f ns/c  %s' % (r['name'], r['candidates'], r['regions'], r['ms'], r['candidates_per_s'], r['ns_per_candidate'], r['kernel_ms']))

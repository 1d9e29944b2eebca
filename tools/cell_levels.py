"""Per-level profile of one sweep cell: python tools/cell_levels.py nx,nt,m [seed]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench, shape_sweep
from ppopt_amd import problem_generator as pg
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
nx, nt, mm = (int(v) for v in sys.argv[1].split(','))
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
prog = bench.program_from_data(pg.generate_mpqp_data(nx, nt, mm, seed))
ml = shape_sweep.explore(prog)[0]
m.solve(prog, max_levels=ml)
prof = []
m.solve(prog, max_levels=ml, profile=prof)
for p in prof:
    if p['depth'] > 0:
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items() if k in ('k', 'candidates', 'status', 'regions', 'ms_kkt', 'ms_theta', 'ms_x', 'ms_xq', 'ms_xq_thread', 'ms_region2', 'n_x_items', 'n_xq_items', 'n_xq_thread', 'ms_wall', 'xtheta_fallbacks', 'ms_verdict', 'ms_region', 'ms_children', 'n_x1', 'ms_x1', 'ms_x_plan', 'lp_pivots', 'xq_pivots', 'n_theta_items', 'n_opt', 'xtheta_lps')})

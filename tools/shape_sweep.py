"""Shape sweep over the north-star class (n_x <= 20, n_theta <= 10): candidate active sets per second of the combinatorial path for
random mpQPs generate_mpqp_data(n_x, n_theta, m, seed) with n_x in {6, 10, 14, 20}, n_theta in {2, 6, 10}, m in {n_x, 2 n_x, 3 n_x},
two mpLPs, BASELINE config 5 (control allocation) and config 1 (transport mpLP).

    python bench.py --sweep                 (one JSON line)        or        python tools/shape_sweep.py [out.json]

Every program is solved to full depth (the reference's loop: max(n_x, n_theta) - n_eq levels) unless its tree passes 10^7 candidates:
then to the deepest level that keeps the total below that (the row says so: `full_depth`).  Per cell: shape after the presolve, levels,
candidates, regions, best of three solves, candidates/s, the kernels' HIP-event times of that solve, and which engine ran
(`rows_theta` / `rows_x` > 64: two tableau rows per lane; deepest cardinality > 8 inequality rows: the KKT solve inside k_theta2 instead
of the one-thread kernel).  The minimum over the class is what the north star's ">= 1e6 candidates/s" is held against.
"""
import json
import os
import sys
import time
import warnings

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BUDGET = 10_000_000       # candidates per solve
FRONTIER_CAP = 6_000_000  # a single level larger than this ends the cell's tree


def explore(prog, device=0):
    """levels of the tree (level by level on the engine) until the budget: returns (max_levels or None for full depth, candidates)"""
    eng = prog.engine(device)
    max_depth = max(prog.num_x(), prog.num_t()) - len(prog.equality_indices)
    eng.pruned_clear()
    eng.frontier_root()
    total, levels = 0, 0
    full = True
    for depth in range(max_depth):
        n = eng.frontier_info()[0] if hasattr(eng, 'frontier_info') else len(eng.frontier_get())
        if total + n > BUDGET or n > FRONTIER_CAP:
            full = False
            break
        gen = depth + 1 != max_depth
        st = eng.level_run(gen)
        total += int(st.n)
        levels += 1
        if not gen or st.n_children == 0:
            break
        eng.frontier_advance()
    return (None if full else levels), total, levels, max_depth


def cell(name, d, device=0, reps=3):
    import bench
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prog = bench.program_from_data(d, device)
    nx, nt, nc, ntc, ne = prog.num_x(), prog.num_t(), prog.num_constraints(), int(prog.A_t.shape[0]), len(prog.equality_indices)
    ml, _, levels, max_depth = explore(prog, device)
    row = {'name': name, 'n_x': nx, 'n_theta': nt, 'n_c': nc, 'n_eq': ne, 'n_tc': ntc, 'mplp': d['Q'] is None,
           'rows_theta': nc - ne + ntc, 'rows_x': nc + ntc - (nx + nt), 'levels': levels, 'full_depth': ml is None, 'max_depth': max_depth}
    if levels == 0:
        row['skipped'] = 'the first level alone passes the budget'
        return row
    m.solve(prog, device=device, max_levels=ml)
    best, prof, nreg = float('inf'), None, 0
    for _ in range(reps):
        pr = []
        t = time.perf_counter()
        sol = m.solve(prog, device=device, max_levels=ml, profile=pr)
        dt = time.perf_counter() - t
        if dt < best:
            best, prof, nreg = dt, pr, len(sol.critical_regions)
        del sol
    lv = [p for p in prof if p['depth'] > 0]
    cands = sum(p['candidates'] for p in prof)
    row.update({'candidates': cands, 'regions': nreg, 'ms': 1e3 * best, 'candidates_per_s': cands / best, 'ns_per_candidate': 1e9 * best / max(cands, 1),
                'largest_level': max(p['candidates'] for p in lv), 'deepest_k': max(p['k'] for p in lv),
                'two_rows_per_lane': {'theta_region': row['rows_theta'] > 64, 'x': row['rows_x'] > 64},
                'kernel_ms': {key: round(sum(p.get(key, 0.0) for p in lv), 3) for key in ('ms_kkt', 'ms_theta', 'ms_x', 'ms_xq', 'ms_xq_thread', 'ms_region2')},
                'stage_ms': {key: round(sum(p.get(key, 0.0) for p in lv), 3) for key in ('ms_verdict', 'ms_region', 'ms_children')},
                # (round 6) the engine from the handle itself (mpc_engine_kind), and the heavy kernels by their HIP-event time over all levels
                **prog.engine(device, closed=True).engine_kind(),
                'kernels_by_time': sorted(((key[3:], round(sum(p.get(key, 0.0) for p in lv), 3)) for key in
                                           ('ms_kkt', 'ms_theta', 'ms_x', 'ms_xq', 'ms_xq_thread', 'ms_x1', 'ms_region2', 'ms_children')), key=lambda kv: -kv[1])})
    row['kkt_in_theta_kernel'] = bool(row['deepest_k'] - ne > row['kkt_thread_max_rows'])      # levels whose KKT systems are solved wavefront-wide inside k_theta2
    prog.release_engine()
    return row


def run(device=0):
    from ppopt_amd import problem_generator as pg
    rows = []
    for nx in (6, 10, 14, 20):
        for nt in (2, 6, 10):
            for mult in (1, 2, 3):
                mm = mult * nx
                rows.append(cell(f'mpqp_{nx}_{nt}_{mm}', pg.generate_mpqp_data(nx, nt, mm, 7), device))
                print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    for (nx, nt, mm) in ((8, 4, 16), (12, 6, 24)):
        d = pg.generate_mpqp_data(nx, nt, mm, 11)
        d['Q'] = None
        rows.append(cell(f'mplp_{nx}_{nt}_{mm}', d, device))
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    rows.append(cell('c5_control_allocation', pg.control_allocation_data(), device))
    rows.append(cell('c1_transport_mplp', pg.transport_mplp_data(), device))
    done = [r for r in rows if 'candidates_per_s' in r]
    big = [r for r in done if r['candidates'] >= 100_000]
    out = {'metric': 'candidate active-sets checked/sec per shape (combinatorial path, one MI355X)', 'budget_candidates': BUDGET, 'cells': rows,
           'min_candidates_per_s': min(r['candidates_per_s'] for r in done), 'min_cell': min(done, key=lambda r: r['candidates_per_s'])['name'],
           'min_candidates_per_s_cells_over_1e5_candidates': (min(r['candidates_per_s'] for r in big) if big else None),
           'min_cell_over_1e5_candidates': (min(big, key=lambda r: r['candidates_per_s'])['name'] if big else None),
           'north_star_target': 1e6,
           'note': 'small trees are bound by launch latency (about 0.1 ms per level whatever its size): cells under 1e5 candidates measure that, not the kernels'}
    return out


if __name__ == '__main__':
    res = run()
    txt = json.dumps(res)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(txt + '\n')
    print(txt)

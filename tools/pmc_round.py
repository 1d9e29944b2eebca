"""Counter record of a round: turns the rocprofv3 --pmc passes of one bench command (and of the calibration micro-kernels) into
profiles/<tag>_pmc.json (tag: environment PMC_TAG, default r04), which bench.py reads for roofline.traffic and for the VALU-issue figures of the simplex kernels.

    python tools/pmc_round.py calib <fetch.db> <write.db>
    python tools/pmc_round.py bench <workload> <solves> fetch=<db> write=<db> [sq=<db>] [f64=<db>]

Calibration (tools/calib/pmc_calib.hip): every micro-kernel moves a known number of bytes over a 2 GiB buffer;
factor = bytes the pattern must move over HBM / (counter x 1024).  For the strided pattern (one 8-byte access per 64-byte
sector) the bytes "that must move" are taken as 64 per touched sector; the counter's reading per sector is stored too.
Per kernel the bench passes apply:  k_xq -> a mix of the coalesced 8-byte factor (columns) and the sector factor (rows), weighted
by the vectors it reads (one column of 63 entries and one row of n_col entries per iteration);  every other kernel -> the
coalesced factors (16 B/lane for k_x2's double2 record copies, 8 B/lane otherwise are equal if the calibration says so).
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md).
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'profiles', os.environ.get('PMC_TAG', 'r06') + '_pmc.json')
LEVELS = {'c4': 5, 'c3': 4, 'c2': 5}
HEAVY = ('k_theta2', 'k_x2', 'k_x1', 'k_region2', 'k_xq_grouped', 'k_xq_thread', 'k_xq', 'k_kkt_thread', 'k_level_small')


def load():
    return json.load(open(OUT)) if os.path.exists(OUT) else {}


def save(d):
    json.dump(d, open(OUT, 'w'), indent=1, sort_keys=True)


def per_kernel(db, counters=None):
    """{kernel short name: {counter: [dispatches, sum]}}, {counter: total over all kernels}"""
    cur = sqlite3.connect(db).cursor()
    out, total = {}, {}
    for name, ctr, n, v in cur.execute('select kernel_name, counter_name, count(*), sum(value) from counters_collection '
                                       'group by kernel_name, counter_name'):
        if counters and ctr not in counters:
            continue
        total[ctr] = total.get(ctr, 0.0) + v
        short = None
        for key in HEAVY:
            if 'mpc::' + key + '<' in name or 'mpc::' + key + '(' in name or name.startswith(key):
                short = key
                break
        if short is None and name.startswith('calib_'):
            short = name.split('(')[0]
        if short is None:
            continue
        a = out.setdefault(short, {}).setdefault(ctr, [0, 0.0])
        a[0] += n
        a[1] += v
    return out, total


def calib(fetch_db, write_db):
    f, _ = per_kernel(fetch_db, ('FETCH_SIZE',))
    w, _ = per_kernel(write_db, ('WRITE_SIZE',))
    buf = float(2 << 30)
    n_sec = (2 << 30) // 8 // 63
    res = {}
    for name, table, ctr in (('calib_read16', f, 'FETCH_SIZE'), ('calib_read8', f, 'FETCH_SIZE'), ('calib_sector8', f, 'FETCH_SIZE'),
                             ('calib_write16', w, 'WRITE_SIZE'), ('calib_write8', w, 'WRITE_SIZE'), ('calib_wsector8', w, 'WRITE_SIZE')):
        n, v = table.get(name, {}).get(ctr, [0, 0.0])
        if not n:
            continue
        counted = 1024.0 * v / n                      # bytes the counter reports per dispatch
        must = 64.0 * n_sec if 'sector' in name else buf
        res[name] = {'dispatches': n, 'counter_bytes_per_dispatch': counted, 'bytes_that_must_move': must, 'factor': must / counted}
        if 'sector' in name:
            res[name]['counter_bytes_per_touched_sector'] = counted / n_sec
    d = load()
    d['calibration'] = res
    save(d)
    print(json.dumps(res, indent=1))


def solves_in(db, fallback):
    """solves of the profiled command = dispatches of k_root_frontier (one per solve); bench.py runs more than steps + warm-up solves since
    round 5 (a second and third timed region, hand-over steps), so the count is taken from the trace itself"""
    cur = sqlite3.connect(db).cursor()
    row = cur.execute("select count(distinct dispatch_id) from counters_collection where kernel_name like '%k_root_frontier%'").fetchone()
    return int(row[0]) if row and row[0] else fallback


def bench(wl, solves, dbs):
    solves = solves_in(dbs['fetch'], solves)
    d = load()
    cal = d.get('calibration', {})
    fac = lambda name, default: cal.get(name, {}).get('factor', default)
    f_r16, f_r8, f_rs = fac('calib_read16', 2.0), fac('calib_read8', 2.0), fac('calib_sector8', 2.0)
    f_w16, f_w8 = fac('calib_write16', 1.0), fac('calib_write8', 1.0)
    f, ftot = per_kernel(dbs['fetch'], ('FETCH_SIZE',))
    w, wtot = per_kernel(dbs['write'], ('WRITE_SIZE',))
    kernels = {}
    for key in sorted(set(f) | set(w)):
        launches = solves * (1 if key in ('k_xq', 'k_xq_grouped') else LEVELS[wl])   # the quick test's wavefront kernel runs on the last level only (its thread pass also plans the storing levels)
        # k_xq: per iteration one column (63 coalesced 8-byte entries) and one row (n_col <= 29 strided entries): weights 63 : 29
        # k_xq_thread: every lane walks its own column with 16-byte loads, 64 distinct sectors per wave-level load: the sector factor
        f_fetch = (63 * f_r8 + 29 * f_rs) / 92.0 if key == 'k_xq' else (f_rs if key == 'k_xq_thread' else (f_r16 if key in ('k_x2', 'k_xq_grouped') else f_r8))
        f_write = f_w16 if key == 'k_x2' else f_w8
        fetch = f_fetch * 1024.0 * f.get(key, {}).get('FETCH_SIZE', [0, 0.0])[1]
        write = f_write * 1024.0 * w.get(key, {}).get('WRITE_SIZE', [0, 0.0])[1]
        kernels[key] = {'bytes_per_launch': (fetch + write) / launches, 'fetch_bytes_per_launch': fetch / launches,
                        'write_bytes_per_launch': write / launches, 'dispatches_profiled': f.get(key, {}).get('FETCH_SIZE', [0])[0],
                        'launches_counted': launches, 'fetch_factor': f_fetch, 'write_factor': f_write}
    res = {'solves': solves, 'fetch_bytes_per_step': f_r8 * 1024.0 * ftot.get('FETCH_SIZE', 0.0) / solves,
           'write_bytes_per_step': f_w8 * 1024.0 * wtot.get('WRITE_SIZE', 0.0) / solves, 'kernels': kernels,
           'factors': {'read16': f_r16, 'read8': f_r8, 'sector8': f_rs, 'write16': f_w16, 'write8': f_w8},
           'note': 'per-step totals use the coalesced 8-byte factors for every kernel; per-kernel figures use the factor of the kernel\'s own pattern'}
    res['bytes_per_step'] = res['fetch_bytes_per_step'] + res['write_bytes_per_step']
    for key, k in kernels.items():      # per step (= per solve): independent of how launches are counted
        k['bytes_per_step'] = k['bytes_per_launch'] * k['launches_counted'] / solves
    for tag in ('sq', 'f64'):
        if tag in dbs and os.path.exists(dbs[tag]):
            t, _ = per_kernel(dbs[tag])
            for key, ctrs in t.items():
                launches = solves * (1 if key in ('k_xq', 'k_xq_grouped') else LEVELS[wl])
                k = kernels.setdefault(key, {'launches_counted': launches})
                sq = k.setdefault('sq_per_launch', {})
                st = k.setdefault('sq_per_step', {})
                for c, (n, v) in ctrs.items():
                    sq[c] = v / launches
                    st[c] = v / solves
    d[wl] = res
    save(d)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == '__main__':
    if sys.argv[1] == 'calib':
        calib(sys.argv[2], sys.argv[3])
    else:
        dbs = dict(a.split('=', 1) for a in sys.argv[4:])
        bench(sys.argv[2], int(sys.argv[3]), dbs)

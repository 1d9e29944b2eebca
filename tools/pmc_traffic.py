"""HBM traffic of a bench step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps S --warmup W
--cpu-sample 0 --locate 0 --mi 0 --complete 0`.  Writes profiles/r02_pmc_traffic.json, which bench.py reads for roofline.traffic.

    python tools/pmc_traffic.py <workload> <fetch.db> <write.db> <solves = S + W>

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-byte read requests at 64 bytes, so it is doubled.  `bytes_per_step` sums every kernel dispatch of the run (the
program set-up launches of the first solve included: a few small LP batches) and divides by the number of solves; the five
heavy kernels are also listed per launch as bench.py counts launches (one k_theta2 / main k_x2 / k_region2 launch per BFS
level and solve), so the small extra dispatches of the same kernels (dictionary-only passes, the base-set check) are folded
into their figure.
"""
import json
import os
import sqlite3
import sys

LEVELS = {'c4': 5, 'c3': 4}
HEAVY = ('k_theta2', 'k_x2', 'k_region2', 'k_xq', 'k_kkt_thread')


def sums(db, counter):
    cur = sqlite3.connect(db).cursor()
    out, total = {}, 0.0
    for name, n, v in cur.execute('select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? '
                                  'group by kernel_name', (counter,)):
        total += v
        for key in HEAVY:
            if 'mpc::' + key + '<' in name:
                a = out.setdefault(key, [0, 0.0])
                a[0] += n
                a[1] += v
    return out, total


def main():
    wl, fdb, wdb, solves = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    (f, ftot), (w, wtot) = sums(fdb, 'FETCH_SIZE'), sums(wdb, 'WRITE_SIZE')
    kernels = {}
    for key in sorted(set(f) | set(w)):
        launches = solves * (1 if key == 'k_xq' else LEVELS[wl])   # the quick test runs on the last level only
        fetch = 2.0 * 1024.0 * f.get(key, [0, 0.0])[1]
        write = 1024.0 * w.get(key, [0, 0.0])[1]
        kernels[key] = {'bytes_per_launch': (fetch + write) / launches, 'fetch_bytes_per_launch': fetch / launches,
                        'write_bytes_per_launch': write / launches, 'dispatches_profiled': f.get(key, [0])[0],
                        'launches_counted': launches, 'fetch_size_doubled': True}
    res = {'solves': solves, 'bytes_per_step': (2.0 * 1024.0 * ftot + 1024.0 * wtot) / solves,
           'fetch_bytes_per_step': 2.0 * 1024.0 * ftot / solves, 'write_bytes_per_step': 1024.0 * wtot / solves,
           'fetch_size_doubled': True, 'kernels': kernels}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r02_pmc_traffic.json')
    allw = json.load(open(path)) if os.path.exists(path) else {}
    allw[wl] = res
    json.dump(allw, open(path, 'w'), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()

"""Summarises a rocprofv3 (ROCm 7.2 rocpd SQLite) result: per-kernel launch count / total / average duration, and the
per-kernel sums of any PMC counters collected.  Usage: python tools/rocpd_summary.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
    rows = cur.execute('select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) '
                       'from kernels group by name order by 3 desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    out.write('kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent\n')
    for name, n, tot, avg, mn, mx in rows:
        out.write(f'"{name}",{n},{tot / 1e6:.4f},{avg / 1e6:.5f},{mn / 1e6:.5f},{mx / 1e6:.5f},{100.0 * tot / total:.2f}\n')
    try:
        pm = cur.execute('select kernel_name, counter_name, count(*), sum(value) from counters_collection '
                         'group by kernel_name, counter_name order by 4 desc').fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        out.write('\nkernel,counter,dispatches,sum\n')
        for k, c, n, v in pm:
            out.write(f'"{k}",{c},{n},{v}\n')
    if out is not sys.stdout:
        out.close()


if __name__ == '__main__':
    main()

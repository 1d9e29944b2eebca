import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute('pragma table_info(regions)')]
rows = list(cur.execute('select name, start, end from regions order by start'))
# last solve: take the last 40% of the trace
t_lo = rows[int(len(rows) * 0.8)][1]
sel = [(n, s, e) for n, s, e in rows if s >= t_lo]
span = sel[-1][2] - sel[0][1]
agg = {}
for n, s, e in sel:
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
print('window ms', span / 1e6, 'calls', len(sel))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f'{n:40s} calls {c:6d} total ms {t/1e6:8.3f} avg us {t/c/1e3:7.1f}')

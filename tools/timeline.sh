# Kernel timeline of the LAST solve of the bench command (run on the GPU box): start offset, duration, gap to the previous kernel end, queue, name
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; wl=${1:-c4}
rm -rf /tmp/pt
(cd $R && rocprofv3 --kernel-trace -d /tmp/pt -o run -- python3 bench.py --workload $wl --steps 6 --warmup 3 --cpu-sample 0 --locate 0 --mi 0 --complete 0 > /tmp/pt.log 2>&1)
python3 - <<PY
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('/tmp/pt/**/*.db',recursive=True)[0]); cur=db.cursor()
cols=[r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol='queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows=cur.execute('select start,end,name%s from kernels order by start' % ((','+qcol) if qcol else '')).fetchall()
roots=[r[0] for r in rows if 'k_root_frontier' in r[2]]
t0=roots[-2]; t1=roots[-1]
rows=[r for r in rows if t0<=r[0]<t1]
short=lambda n: re.sub(r'\(.*','',n).replace('void mpc::','').replace('mpc::','')[:34]
last_end=rows[0][0]
for r in rows:
    s,e,n=r[0],r[1],r[2]
    gap=(s-last_end)/1e3
    print('%9.1f us  dur %8.1f  gap %7.1f  q%-3s %s'%((s-t0)/1e3,(e-s)/1e3,gap,r[3] if qcol else '-',short(n)))
    last_end=max(last_end,e)
PY

# Host calls and kernels of the LAST solve of the bench command on one time axis (run on the GPU box): bash tools/host_timeline.sh <workload>
# rocprofv3 --hip-trace --kernel-trace (no counters): which HIP call of the host sits in which gap of the device's timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; wl=${1:-c2}
rm -rf /tmp/ph
(cd $R && rocprofv3 --hip-trace --kernel-trace -d /tmp/ph -o run -- python3 bench.py --workload $wl --steps 6 --warmup 3 --cpu-sample 0 --locate 0 --mi 0 --complete 0 > /tmp/ph.log 2>&1)
python3 - <<PY
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('/tmp/ph/**/*.db',recursive=True)[0]); cur=db.cursor()
ks=cur.execute('select start,end,name from kernels order by start').fetchall()
roots=[r[0] for r in ks if 'k_root_frontier' in r[2]]
t0=roots[-2]-30000; t1=roots[-1]-30000
cols=[r[1] for r in cur.execute("pragma table_info(regions)").fetchall()]
tcol='tid' if 'tid' in cols else None
rs=cur.execute('select start,end,name%s from regions order by start' % ((','+tcol) if tcol else '')).fetchall()
short=lambda n: re.sub(r'\(.*','',n).replace('void mpc::','').replace('mpc::','')[:40]
ev=[(s,'K',e-s,short(n),'') for s,e,n in ks if t0<=s<t1]+[(r[0],'H',r[1]-r[0],r[2],r[3] if tcol else '') for r in rs if t0<=r[0]<t1]
ev.sort()
for s,kind,d,n,t in ev:
    if kind=='K': print('%9.1f                                        GPU %7.1f  %s'%((s-t0)/1e3,d/1e3,n))
    else: print('%9.1f  host %7.1f  %-28s %s'%((s-t0)/1e3,d/1e3,n[:28],t))
PY

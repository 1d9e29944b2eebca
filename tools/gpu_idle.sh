# GPU timeline of the bench command: union of kernel intervals, idle time inside the timed steps, gaps by length (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; wl=${1:-c4}
rm -rf /tmp/pi
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/pi -o run -- python3 bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample 0 --locate 0 --mi 0 --complete 0 > /tmp/pi.log 2>&1)
python3 - <<PY
import sqlite3, glob
db=sqlite3.connect(glob.glob('/tmp/pi/**/*.db',recursive=True)[0]); cur=db.cursor()
rows=cur.execute('select start,end,name from kernels order by start').fetchall()
# the timed steps are the last 10 solves: split the timeline at k_root_frontier launches
roots=[s for s,e,n in rows if 'k_root_frontier' in n]
t0=roots[-10]; rows=[r for r in rows if r[0]>=t0]
iv=[]; 
for s,e,n in rows:
    if iv and s<=iv[-1][1]: iv[-1][1]=max(iv[-1][1],e)
    else: iv.append([s,e])
union=sum(e-s for s,e in iv); span=iv[-1][1]-iv[0][0]
gaps=[iv[i+1][0]-iv[i][1] for i in range(len(iv)-1)]
import collections
b=collections.Counter()
for g in gaps:
    k='<5us' if g<5e3 else '<15us' if g<15e3 else '<30us' if g<30e3 else '<100us' if g<100e3 else '>=100us'
    b[k]+=g
import re
short=lambda n: re.sub(r'\(.*','',n).replace('void mpc::','').replace('mpc::','')[:28]
# which kernel pairs the gaps sit between (gaps of the merged timeline: map interval ends back to kernels)
ends={}
for s_,e_,n_ in rows: ends[e_]=n_
starts={}
for s_,e_,n_ in rows: starts.setdefault(s_,n_)
pairs=collections.Counter(); pt=collections.Counter()
for i in range(len(iv)-1):
    g=iv[i+1][0]-iv[i][1]
    if g>=10e3:
        key=(short(ends.get(iv[i][1],'?')), short(starts.get(iv[i+1][0],'?')))
        pairs[key]+=1; pt[key]+=g
for key,v in sorted(pt.items(), key=lambda kv:-kv[1])[:18]:
    print('  gap after %-28s before %-28s : %5.1f per solve, %6.1f us each, %.3f ms per solve'%(key[0],key[1],pairs[key]/10,v/pairs[key]/1e3,v/1e7))
print('per solve: span %.3f ms, busy (union) %.3f ms, idle %.3f ms; idle by gap length (ms per solve):'%(span/1e7,union/1e7,(span-union)/1e7), {k:round(v/1e7,3) for k,v in b.items()}, 'gaps per solve', len(gaps)/10)
PY

# kernel trace of bench.py --workload $1 with and without the small path: per-kernel calls / avg duration (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; wl=${1:-c2}
O=$R/gpurun_out/prof3; mkdir -p $O
ARGS="bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample 0 --locate 0 --mi 0 --complete 0"
rm -rf /tmp/ps0 /tmp/ps1
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/ps0 -o run -- python3 $ARGS > $O/ps0_$wl.log 2>&1)
export MPC_NO_SMALLPATH=1
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/ps1 -o run -- python3 $ARGS > $O/ps1_$wl.log 2>&1)
unset MPC_NO_SMALLPATH
python3 $R/tools/rocpd_summary.py $(find /tmp/ps0 -name "*.db" | head -1) $O/small_${wl}_kernel_stats.csv > /dev/null
python3 $R/tools/rocpd_summary.py $(find /tmp/ps1 -name "*.db" | head -1) $O/classic_${wl}_kernel_stats.csv > /dev/null
python3 - <<PY
import sqlite3,sys
for tag,d in (('small','/tmp/ps0'),('classic','/tmp/ps1')):
    import glob
    db=sqlite3.connect(glob.glob(d+'/**/*.db',recursive=True)[0]); cur=db.cursor()
    rows=cur.execute('select start,end from kernels order by start').fetchall()
    busy=sum(e-s for s,e in rows); span=rows[-1][1]-rows[0][0]
    gaps=sorted([rows[i+1][0]-rows[i][1] for i in range(len(rows)-1)])
    print(tag, 'kernels',len(rows),'busy ms %.2f'%(busy/1e6),'span ms %.1f'%(span/1e6),'median gap us %.2f'%(gaps[len(gaps)//2]/1e3))
PY

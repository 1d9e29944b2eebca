# kernel trace and HIP API trace of the batched mixed-integer enumeration (tools/mi_batch.py, MI_BATCH_REPS solves) -- run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
export MI_BATCH_ONLY=1
rm -rf /tmp/pmb0 /tmp/pmb1
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/pmb0 -o run -- python3 tools/mi_batch.py > $O/mi_batch_ktrace.log 2>&1)
python3 $R/tools/rocpd_summary.py $(find /tmp/pmb0 -name "*.db" | head -1) $O/mi_batch_kernel_stats.csv | head -30
(cd $R && rocprofv3 --hip-trace --stats -d /tmp/pmb1 -o run -- python3 tools/mi_batch.py > $O/mi_batch_hiptrace.log 2>&1)
tail -5 $O/mi_batch_hiptrace.log
python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/pmb1/**/*.db', recursive=True)[0]); cur = db.cursor()
try:
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from regions group by name order by 3 desc limit 25").fetchall()
    for r in rows: print('%-40s calls %7d total %9.2f ms avg %9.2f us' % r)
except Exception as e:
    print('regions query failed', e)
PY

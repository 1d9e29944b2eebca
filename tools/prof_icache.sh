# Instruction-cache counters of the bench command, per kernel (run on the GPU box):  bash tools/prof_icache.sh <workload> [extra env as VAR=VALUE ...]
# One --pmc pass (no trace domains beside it); the program goes directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; wl=${1:-c4}; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/icache; mkdir -p $O
rm -rf /tmp/pic
(cd $R && rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pic -o run -- python3 bench.py --workload $wl --steps 6 --warmup 3 --cpu-sample 0 --locate 0 --mi 0 --complete 0 > $O/pic_$wl.log 2>&1)
D=$(find /tmp/pic -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $D $O/icache_${wl}.csv > /dev/null
python3 - <<PY
import re, collections
tab=collections.defaultdict(dict)
sect=False
for line in open('$O/icache_${wl}.csv'):
    if line.startswith('kernel,counter'): sect=True; continue
    if not sect or not line.strip(): continue
    m=re.match(r'"(.*)",([A-Z_0-9]+),(\d+),([0-9.e+]+)', line)
    if m:
        name=re.sub(r'\(.*','',m.group(1)).replace('void mpc::','')[:40]
        tab[name][m.group(2)]=float(m.group(4)); tab[name]['n']=int(m.group(3))
print('%-40s %6s %12s %12s %8s %12s %10s' % ('kernel','disp','icache_req','misses','miss %','wave_cycles','req/wcyc'))
for k,v in sorted(tab.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',0))[:14]:
    rq=v.get('SQC_ICACHE_REQ',0); ms=v.get('SQC_ICACHE_MISSES',0); wc=v.get('SQ_WAVE_CYCLES',0)
    print('%-40s %6d %12.3g %12.3g %8.2f %12.3g %10.4f  dup %.3g ifetch %.3g busy %.3g' % (k, v['n'], rq, ms, 100*ms/max(rq,1), wc, rq/max(wc,1), v.get('SQC_ICACHE_MISSES_DUPLICATE',0), v.get('SQ_IFETCH',0), v.get('SQ_BUSY_CYCLES',0)))
PY

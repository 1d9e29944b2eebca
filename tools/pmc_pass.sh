# One --pmc pass over the bench command (GPU box):  bash tools/pmc_pass.sh <tag> "<counters>" [VAR=VALUE ...]
# writes gpurun_out/pmc/<tag>.csv (tools/rocpd_summary.py) and prints the sums of the heavy kernels.  The program goes directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; ctrs=$2; shift 2
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/pmc; mkdir -p $O
rm -rf /tmp/pp_$tag
(cd $R && rocprofv3 --pmc $ctrs -d /tmp/pp_$tag -o run -- python3 bench.py --workload ${WL:-c4} --steps 6 --warmup 3 --cpu-sample 0 --locate 0 --mi 0 --complete 0 > $O/$tag.log 2>&1)
D=$(find /tmp/pp_$tag -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $D $O/$tag.csv > /dev/null
python3 - <<PY
import re, collections
tab=collections.defaultdict(dict); sect=False
for line in open('$O/$tag.csv'):
    if line.startswith('kernel,counter'): sect=True; continue
    if not sect or not line.strip(): continue
    m=re.match(r'"(.*)",([A-Za-z_0-9]+),(\\d+),([0-9.e+]+)', line)
    if m:
        name=re.sub(r'\\(.*','',m.group(1)).replace('void mpc::','').replace('mpc::','')[:28]
        tab[name][m.group(2)]=float(m.group(4)); tab[name]['n']=int(m.group(3))
cs=sorted({c for v in tab.values() for c in v if c!='n'})
print('$tag', ' '.join('$@'.split()))
print('%-28s %5s ' % ('kernel','disp') + ' '.join('%14s' % c[-14:] for c in cs))
for k,v in sorted(tab.items(), key=lambda kv:-max([x for c,x in kv[1].items() if c!='n'] or [0]))[:6]:
    print('%-28s %5d ' % (k, v['n']) + ' '.join('%14.4g' % v.get(c,0) for c in cs))
PY

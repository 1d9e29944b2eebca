# usage: bash tools/prof_one.sh <tag> [bench args...]   (environment variables pass through) -> gpurun_out/r2/kstats_<tag>.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/prof_$tag
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python3 bench.py --steps 3 --warmup 2 --cpu-sample 0 --locate 0 --mi 0 "$@" > /tmp/prof_$tag.log 2>&1)
DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
mkdir -p $R/gpurun_out/r2
python3 $R/tools/rocpd_summary.py $DB $R/gpurun_out/r2/kstats_$tag.csv > /dev/null 2>&1
python3 - <<PY
import csv
print("$tag")
for row in list(csv.reader(open("$R/gpurun_out/r2/kstats_$tag.csv")))[1:8]:
    print("  ", row[0][:44].ljust(46), row[1:6])
PY

# kernel stats of a complete graph solve (run on the GPU box): bash tools/prof_graph.sh [c4|c3]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
wl=${1:-c4}
rm -rf /tmp/pg
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/pg -o run -- python3 tools/graph_split.py $wl > /tmp/pg.log 2>&1)
DB=$(find /tmp/pg -name "*.db" | head -1)
mkdir -p $R/gpurun_out/prof
python3 $R/tools/rocpd_summary.py $DB $R/gpurun_out/prof/r02_graph_${wl}_kernel_stats.csv > /dev/null 2>&1
python3 - <<PY
import csv
for row in list(csv.reader(open("$R/gpurun_out/prof/r02_graph_${wl}_kernel_stats.csv")))[1:16]:
    print("  ", row[0][:52].ljust(54), row[1:6])
PY

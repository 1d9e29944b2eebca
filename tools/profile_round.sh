# Round profile of the bench command (run on the GPU box):  bash tools/profile_round.sh <workload> <tag>
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md), summaries under gpurun_out/prof/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
wl=${1:-c4}; tag=${2:-r02}; steps=${3:-20}; warm=${4:-5}
mkdir -p $R/gpurun_out/prof
ARGS="bench.py --workload $wl --steps $steps --warmup $warm --cpu-sample 0 --locate 0 --mi 0 --complete 0"
rm -rf /tmp/pk /tmp/pf /tmp/pw /tmp/psq
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/pk -o run -- python3 $ARGS > /tmp/pk.log 2>&1)
(cd $R && rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o run -- python3 $ARGS > /tmp/pf.log 2>&1)
(cd $R && rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o run -- python3 $ARGS > /tmp/pw.log 2>&1)
K=$(find /tmp/pk -name "*.db" | head -1); F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $K $R/gpurun_out/prof/${tag}_${wl}_kernel_stats.csv > /dev/null
python3 $R/tools/rocpd_summary.py $F $R/gpurun_out/prof/${tag}_${wl}_pmc_fetch_size.csv > /dev/null
python3 $R/tools/rocpd_summary.py $W $R/gpurun_out/prof/${tag}_${wl}_pmc_write_size.csv > /dev/null
(cd $R && python3 tools/pmc_traffic.py $wl $F $W $((steps + warm)) > $R/gpurun_out/prof/${tag}_${wl}_traffic.txt && cp profiles/r02_pmc_traffic.json gpurun_out/prof/)
head -12 $R/gpurun_out/prof/${tag}_${wl}_kernel_stats.csv | cut -c1-50,180-330

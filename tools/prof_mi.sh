# kernel trace of the mixed-integer enumeration (bench's extra workload, 8 sub-programs in flight): per-kernel totals (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/pm -o run -- python3 tools/mi_queues.py ${1:-1} > /tmp/pm.log 2>&1)
mkdir -p $R/gpurun_out/prof3
python3 $R/tools/rocpd_summary.py $(find /tmp/pm -name "*.db" | head -1) $R/gpurun_out/prof3/r03_mi_kernel_stats.csv > /dev/null
tail -2 /tmp/pm.log

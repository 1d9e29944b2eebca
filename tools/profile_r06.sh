# Round-6 records (run on the GPU box): bash tools/profile_r06.sh <part>   part 1: bench lines, level walls, timelines, the k > 8 level;  part 2: c3 / c2 profile passes, ranks on one GPU, the shape sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6prof; mkdir -p $O; cd $R
if [ "$1" = "1" ]; then
  python bench.py > $O/r06_bench_c4.json 2> $O/bench_c4.err
  for w in c3 c2 c2x20 c5 c1; do python bench.py --workload $w --cpu-sample 0 --mi 0 --complete 0 --locate 0 > $O/r06_bench_$w.json 2> $O/bench_$w.err; done
  for w in c4 c3 c2 c5; do python tools/level_walls.py $w 15 > $O/r06_${w}_level_walls.log 2>/dev/null; python tools/solve_time.py $w 40 >> $O/r06_${w}_level_walls.log 2>/dev/null; MPC_X1_DEFER=0 python tools/solve_time.py $w 40 2>/dev/null | sed 's/^/MPC_X1_DEFER=0 /' >> $O/r06_${w}_level_walls.log; done
  bash tools/timeline.sh c4 > $O/r06_c4_timeline.log 2>&1
  bash tools/timeline.sh c2 > $O/r06_c2_timeline.log 2>&1
  bash tools/gpu_idle.sh c4 > $O/r06_c4_gpu_idle.log 2>&1
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/k9 /tmp/k9p
  (cd $R && rocprofv3 --kernel-trace --stats -d /tmp/k9 -o run -- python3 tools/dbg_levels.py 10 6 20 > $O/r06_k9_levels.log 2>&1)
  (cd $R && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/k9p -o run -- python3 tools/dbg_levels.py 10 6 20 > $O/k9p.log 2>&1)
  python3 $R/tools/rocpd_summary.py $(find /tmp/k9 -name "*.db" | head -1) $O/r06_k9_kernel_stats.csv > /dev/null
  python3 $R/tools/rocpd_summary.py $(find /tmp/k9p -name "*.db" | head -1) $O/r06_k9_pmc_sq.csv > /dev/null
else
  bash tools/profile_round3.sh c3 20 5 r06 > $O/prof_c3.log 2>&1
  bash tools/profile_round3.sh c2 20 5 r06 > $O/prof_c2.log 2>&1
  cp $R/gpurun_out/prof3/r06_c3_* $R/gpurun_out/prof3/r06_c2_* $R/gpurun_out/prof3/r06_pmc.json $O/ 2>/dev/null
  cd $R
  python tools/ranks_one_gpu.py c4 2 4 8 > $O/r06_ranks_c4.log 2>&1; cp gpurun_out/r6/ranks_c4.json $O/r06_ranks_c4.json
  python tools/ranks_one_gpu.py c3 2 4 8 > $O/r06_ranks_c3.log 2>&1; cp gpurun_out/r6/ranks_c3.json $O/r06_ranks_c3.json
  python bench.py --sweep > $O/r06_sweep.json 2> $O/sweep.err
fi
ls $O

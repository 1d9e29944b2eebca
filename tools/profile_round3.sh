# Profile of the bench command (run on the GPU box):  bash tools/profile_round3.sh <workload> [steps] [warmup] [tag]   (tag: r03, r04, r05 ...; default r05)
# 1. calibration of FETCH_SIZE / WRITE_SIZE on known byte counts (tools/calib/pmc_calib, two passes)
# 2. kernel trace + stats of `bench.py --workload W --steps S --warmup U --cpu-sample 0 --locate 0 --mi 0 --complete 0`
# 3. FETCH_SIZE, WRITE_SIZE, SQ and (if the counters exist) VALU-F64 instruction counters, each in its own --pmc pass
# Under rocprofv3 the program goes directly after `--` (python3 bench.py ... / the binary): no wrapper, no env, no shell.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
wl=${1:-c4}; steps=${2:-20}; warm=${3:-5}; tag=${4:-r06}; export PMC_TAG=$tag
O=$R/gpurun_out/prof3; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
if [ ! -f $O/calib_done ]; then
  rm -rf /tmp/cf /tmp/cw
  (cd $R && rocprofv3 --pmc FETCH_SIZE -d /tmp/cf -o run -- ./tools/calib/pmc_calib > $O/calib_fetch.log 2>&1)
  (cd $R && rocprofv3 --pmc WRITE_SIZE -d /tmp/cw -o run -- ./tools/calib/pmc_calib > $O/calib_write.log 2>&1)
  CF=$(find /tmp/cf -name "*.db" | head -1); CW=$(find /tmp/cw -name "*.db" | head -1)
  python3 $R/tools/rocpd_summary.py $CF $O/${tag}_calib_fetch.csv > /dev/null
  python3 $R/tools/rocpd_summary.py $CW $O/${tag}_calib_write.csv > /dev/null
  (cd $R && python3 tools/pmc_round.py calib $CF $CW > $O/${tag}_calibration.txt 2>&1) && touch $O/calib_done
fi
ARGS="bench.py --workload $wl --steps $steps --warmup $warm --cpu-sample 0 --locate 0 --mi 0 --complete 0 --events-in-value"   # (one timed region under the profiler: steps + warm-up solves)
rm -rf /tmp/pk /tmp/pf /tmp/pw /tmp/psq /tmp/p64
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/pk -o run -- python3 $ARGS > $O/pk_$wl.log 2>&1)
(cd $R && rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o run -- python3 $ARGS > $O/pf_$wl.log 2>&1)
(cd $R && rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o run -- python3 $ARGS > $O/pw_$wl.log 2>&1)
(cd $R && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/psq -o run -- python3 $ARGS > $O/psq_$wl.log 2>&1)
(cd $R && rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d /tmp/p64 -o run -- python3 $ARGS > $O/p64_$wl.log 2>&1)
K=$(find /tmp/pk -name "*.db" | head -1); F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
S=$(find /tmp/psq -name "*.db" | head -1); D=$(find /tmp/p64 -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $K $O/${tag}_${wl}_kernel_stats.csv > /dev/null
python3 $R/tools/rocpd_summary.py $F $O/${tag}_${wl}_pmc_fetch_size.csv > /dev/null
python3 $R/tools/rocpd_summary.py $W $O/${tag}_${wl}_pmc_write_size.csv > /dev/null
[ -n "$S" ] && python3 $R/tools/rocpd_summary.py $S $O/${tag}_${wl}_pmc_sq.csv > /dev/null
[ -n "$D" ] && python3 $R/tools/rocpd_summary.py $D $O/${tag}_${wl}_pmc_f64.csv > /dev/null
(cd $R && python3 tools/pmc_round.py bench $wl $((steps + warm)) fetch=$F write=$W ${S:+sq=$S} ${D:+f64=$D} > $O/${tag}_${wl}_pmc.txt 2>&1; cp profiles/${tag}_pmc.json $O/)
head -14 $O/${tag}_${wl}_kernel_stats.csv | cut -c1-60,180-330

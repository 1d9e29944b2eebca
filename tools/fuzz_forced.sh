# Fuzz with the large-level paths forced onto every level (one-thread quick test, one-step plans, classic path with the spread KKT form, the
# bucketed children scan) and the shared-launch fuzz (GPU box): bash tools/fuzz_forced.sh > gpurun_out/fuzz_forced.log
export MPC_NO_SMALLPATH=1 MPC_XQT_MIN=1 MPC_X1_MIN=1 MPC_PRUNED_BUCKET_MIN=1 MPC_PRUNED_BUCKET_NP=1
for c in mpqp mpqp_eq mplp open; do timeout 1200 python tools/fuzz_scan.py 100 $c 2061 2>&1 | tail -1 | sed "s/^/forced paths: /"; done
unset MPC_NO_SMALLPATH MPC_XQT_MIN MPC_X1_MIN MPC_PRUNED_BUCKET_MIN MPC_PRUNED_BUCKET_NP
timeout 1200 python tools/fuzz_batch.py 300 2062 50 5 2>&1 | tail -2 | sed "s/^/shared launches: /"

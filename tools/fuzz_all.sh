# The round's closing fuzz: every class of tools/fuzz_scan.py against the CPU oracle (GPU box): bash tools/fuzz_all.sh > gpurun_out/fuzz.log
for c in mpqp mplp open mpqp_eq mpc; do timeout 900 python tools/fuzz_scan.py 120 $c 2060 2>&1 | tail -1 | sed "s/^/$c /"; done
timeout 1500 python tools/fuzz_scan.py 40 big 2060 2>&1 | grep -v "^program" | tail -2 | sed "s/^/big /"

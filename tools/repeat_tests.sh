# repeats a test selection until one run fails or hangs (per-test timeout with stack dump); usage: bash tools/repeat_tests.sh N <pytest args>
n=$1; shift
for i in $(seq 1 $n); do
  timeout 900 python -m pytest "$@" -x -q -m gpu --timeout=600 --timeout-method=thread > /tmp/rep_$i.log 2>&1
  rc=$?
  tail -1 /tmp/rep_$i.log
  if [ $rc -ne 0 ]; then echo "RUN $i FAILED rc=$rc"; grep -v "^\s*$" /tmp/rep_$i.log | tail -120; break; fi
done

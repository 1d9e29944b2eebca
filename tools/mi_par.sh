for np in 1 2 4; do
  echo "== $np processes"
  for i in $(seq $np); do python tools/mi_queues.py 8 8 > /tmp/mi_$i.log 2>&1 & done
  wait
  for i in $(seq $np); do tail -1 /tmp/mi_$i.log; done
done

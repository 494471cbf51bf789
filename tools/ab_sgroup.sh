for r in 1 2; do
for G in 1 2 3 4 6; do
  echo "== G=$G"
  IPX_FI_STRIP_GROUP=$G timeout -k 10 120 python tools/bench_stage.py flow 513 3 2>&1 | grep "flow_metric\|flow_iter" || exit 1
done
done

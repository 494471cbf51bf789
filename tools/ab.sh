#!/bin/bash
# A/B timing of the flow stage on one box: tools/ab.sh <libA> <libB> [frames]   (alternating, 2 rounds)
A=$1; B=$2; N=${3:-257}
for r in 1 2; do
  for L in "$A" "$B"; do
    echo "== $L"
    IPX_LIB=$L timeout -k 10 120 python tools/bench_stage.py flow $N 3 2>&1 | grep "flow_metric\|flow_iter\|polyexp\|pyr_level" || exit 1
  done
done

#!/bin/bash
# alternating timing of the flow stage with several builds on one box: tools/ab3.sh <frames> <lib> <lib> ...
N=$1; shift
for r in 1 2; do
  for L in "$@"; do
    echo "== $L"
    IPX_LIB=$L timeout -k 10 120 python tools/bench_stage.py flow $N 3 2>&1 | grep "flow_metric\|flow_iter" || exit 1
  done
done

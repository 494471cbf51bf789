#!/bin/bash
# tools/abn.sh <lib> [<lib> ...]: the flow stage on a whole 1000-frame 1080p chunk with every library in turn, two rounds
for r in 1 2; do for L in "$@"; do echo "== $L"; IPX_LIB=$L timeout -k 10 200 python tools/bench_stage.py flow 1000 3 2>&1 | grep "flow_metric\|flow_iter\|polyexp\|pyr_level" || exit 1; done; done

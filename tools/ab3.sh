#!/bin/bash
# tools/ab3.sh <lib> [<lib> ...]: flow stage timing of several builds on one box, two rounds
for r in 1 2; do for L in "$@"; do echo "== $L"; IPX_LIB=$L timeout -k 10 120 python tools/bench_stage.py flow 257 3 2>&1 | grep "flow_iter\|polyexp\|pyr_level" || exit 1; done; done

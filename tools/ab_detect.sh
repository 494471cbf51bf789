#!/bin/bash
# tools/ab_detect.sh <lib> [<lib> ...]: detect_kernel on a resident 1000-frame 1080p chunk with every library in turn, three rounds
for r in 1 2 3; do for L in "$@"; do echo "== $L"; IPX_LIB=$L timeout -k 10 100 python tools/bench_stage.py detect 1000 3 2>&1 | grep "^detect " || exit 1; done; done

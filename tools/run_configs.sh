#!/bin/bash
# The other BASELINE configurations (never `value`): tools/run_configs.sh <tag>  -> gpurun_out/<tag>_bench_*.json
set -e
cd "$(dirname "$0")/.."
tag=${1:-r02}
timeout -k 10 400 python bench.py --height 2160 --width 3840 --cpu-frames 0 --no-modes --steps 2 > gpurun_out/${tag}_bench_4k.json
echo 4k done
timeout -k 10 300 python bench.py --height 480 --width 640 --frames 300 --cpu-frames 0 --no-modes --steps 5 > gpurun_out/${tag}_bench_640x480.json
echo vga done
timeout -k 10 300 python bench.py --motion --cpu-frames 0 --no-modes > gpurun_out/${tag}_bench_motion.json
echo motion done
timeout -k 10 400 python bench.py --videos 2 --height 2160 --width 3840 --steps 1 --warmup 1 > gpurun_out/${tag}_bench_videos_2x4k.json
echo videos done

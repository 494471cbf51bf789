#!/bin/bash
# Runs on the GPU box (via gpurun): the rocprofv3 evidence behind bench.py's `roofline`.
#   tools/profile.sh <tag>      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# 1. kernel trace + stats of the default bench command; 2./3. the two TCC passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), one chunk each, no other tracing.
set -e
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o ${tag} -- \
    python3 $root/bench.py --cpu-frames 0 --no-modes > $out/${tag}_bench_under_rocprof.log 2>&1
echo "stats pass done"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o ${tag} -- \
    python3 $root/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-modes > $out/${tag}_pmc_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o ${tag} -- \
    python3 $root/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-modes > $out/${tag}_pmc_write.log 2>&1
echo "write pass done"
cd $root
python3 tools/pmc_traffic.py $tag

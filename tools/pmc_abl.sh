#!/bin/bash
# FETCH_SIZE of the flow stage for an ablation build (tools/fi_ablate.py --build <bits>): tools/pmc_abl.sh <bits>
set -e
bits=$1
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp; cd /tmp
export IPX_LIB=$root/image_processing_amd/lib_abl$bits/libipx.so
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/abl${bits}_fetch -o abl -- python3 $root/tools/bench_stage.py flow 513 1 > $out/abl${bits}_fetch.log 2>&1
echo "ablation $bits fetch pass done"

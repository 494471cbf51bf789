#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the flow stage's launches for two workgroup orders of flow_iter_kernel (IPX_FI_STRIP_GROUP)
set -e
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp; cd /tmp
for G in 1 3; do
  IPX_FI_STRIP_GROUP=$G timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/sg${G}_fetch -o sg -- python3 $root/tools/bench_stage.py flow 513 1 > $out/sg${G}_fetch.log 2>&1
  echo "G=$G fetch pass done"
done

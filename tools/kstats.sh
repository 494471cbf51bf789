#!/bin/bash
# per-kernel times of one stage under rocprofv3 (on the GPU box): tools/kstats.sh <tag> <stage> [frames]
tag=$1; stage=$2; n=${3:-257}
root=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o $tag -- python3 $root/tools/bench_stage.py $stage $n 2 > /tmp/ks_$tag.log 2>&1 || { tail -5 /tmp/ks_$tag.log; exit 1; }
f=$(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "anonymous" in n and "at::" not in n:
        print("%-60s calls %4s avg %10.1f us" % (n.split("(anonymous namespace)::")[1].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY

#!/bin/bash
# One PMC pass over one chunk: tools/pmc_one.sh <tag> <counter>[,<counter>...]  (run on the GPU box)
set -e
tag=$1; ctr=$2
root=$(pwd); out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 400 rocprofv3 --pmc ${ctr//,/ } --output-format csv -d $out/${tag}_pmc -o ${tag} -- \
    python3 $root/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-modes > $out/${tag}_pmc.log 2>&1
cd $root
python3 - "$out/${tag}_pmc/${tag}_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "anonymous" not in n or "at::" in n:
        continue
    k = (n.split("(anonymous namespace)::")[1].split("(")[0], r["Counter_Name"])
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print("%-28s %-22s n=%5d avg=%16.1f" % (k, c, n, v / n))
PY

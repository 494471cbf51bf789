#!/bin/bash
# PMC passes over one stage: tools/pmc_stage.sh <tag> <stage> "<counters pass 1>" ["<counters pass 2>" ...]  (on the GPU box)
set -e
tag=$1; stage=$2; shift 2
root=$(pwd); out=$root/gpurun_out; raw=/tmp/ipx_pmc_$tag   # raw counter files stay off gpurun_out (64 MiB limit)
mkdir -p $raw
export TMPDIR=/tmp
cd /tmp
i=0
for ctr in "$@"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $raw/pmc$i -o ${tag} -- \
        python3 $root/tools/bench_stage.py $stage 1000 1 > $out/${tag}_pmc$i.log 2>&1
    python3 - "$raw/pmc$i" <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: [0, 0.0])
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "anonymous" not in n or "at::" in n:
            continue
        k = (n.split("(anonymous namespace)::")[1].split("(")[0][:40], r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print("%-40s %-26s n=%4d avg=%18.1f" % (k, c, n, v / n))
PY
done

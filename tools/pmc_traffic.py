#!/usr/bin/env python3
"""Folds the rocprofv3 outputs of tools/profile.sh into small summaries:

  gpurun_out/<tag>_kernel_stats.csv   per-kernel calls / total / average duration (from --stats)
  gpurun_out/<tag>_traffic.json       HBM bytes per launch per kernel class from the two TCC passes

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte fabric reads as 64 bytes
(MI355X_MICROARCH.md, HBM section), so reads are doubled; the factor is checked in the same run on
kernels whose read volume is known exactly (persistent, magnitude: pure streaming reads).
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
FETCH_FACTOR = 2.0

# kernel symbol -> profiler class name of libipx (api.hip)
CLASSES = [("flow_iter_kernel", "flow_iter"), ("pyr_level_kernel", "pyr_level"), ("pyr_exact_kernel", "pyr_level"), ("pyr3_kernel", "pyr_level"), ("polyexp_kernel", "polyexp"),
           ("magnitude_finish_kernel", "magnitude_finish"),
           ("detect_kernel", "detect"), ("persistent_kernel", "persistent"), ("static_maps_kernel", "static_maps"), ("static_maps16_kernel", "static_maps"),
           ("finalize_kernel", "finalize"), ("ccl_merge", "ccl_merge"), ("ccl_init", "ccl_init")]


def classify(name):
    for sym, cls in CLASSES:
        if sym in name:
            return cls
    return None


def find(tag, sub, pattern):
    hits = glob.glob(os.path.join(OUT, "%s_%s" % (tag, sub), "**", pattern), recursive=True)
    return sorted(hits)


def counter_per_launch(tag, sub, counter):
    acc = {}
    for p in find(tag, sub, "*counter_collection.csv"):
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                cls = classify(row.get("Kernel_Name", ""))
                if cls is None:
                    continue
                a = acc.setdefault(cls, [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    stats = find(tag, "stats", "*kernel_stats.csv")
    if stats:
        dst = os.path.join(OUT, "%s_kernel_stats.csv" % tag)
        with open(stats[0]) as f, open(dst, "w") as g:
            g.write(f.read())
        print("wrote", dst)
    fetch = counter_per_launch(tag, "pmc_fetch", "FETCH_SIZE")
    write = counter_per_launch(tag, "pmc_write", "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, kf = fetch.get(k, (0, 0.0))
        nw, kw = write.get(k, (0, 0.0))
        rd = kf * 1024.0 * FETCH_FACTOR
        wr = kw * 1024.0
        kernels[k] = dict(launches=max(nf, nw), fetch_size_kib_raw=round(kf, 1), write_size_kib_raw=round(kw, 1),
                          hbm_read_bytes_per_launch=round(rd), hbm_write_bytes_per_launch=round(wr),
                          hbm_bytes_per_launch=round(rd + wr))
    log = os.path.join(OUT, "%s_pmc_fetch.log" % tag)
    workload = None
    if os.path.exists(log):
        m = re.search(r'"workload": "([^"]+)"', open(log).read())
        workload = m.group(1) if m else None
    doc = dict(workload=workload, source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0",
               correction="FETCH_SIZE x %.1f (gfx950 tallies 128-B reads as 64 B); WRITE_SIZE as reported" % FETCH_FACTOR,
               kernels=kernels)
    dst = os.path.join(OUT, "%s_traffic.json" % tag)
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", dst)
    for k, v in kernels.items():
        print("%-16s n=%5d read %10.1f MB  write %10.1f MB" % (k, v["launches"], v["hbm_read_bytes_per_launch"] / 1e6,
                                                           v["hbm_write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()

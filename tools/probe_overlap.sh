#!/bin/bash
# Can vector arithmetic from another stream run in the issue slots the marching flow kernel leaves idle?
# (flow_iter_kernel's two workgroups hold 158.7 of a CU's 160 KB of LDS, so only LDS-free work can join them.)
# An LDS-free burn kernel (flow.hip: fi_probe_burn_kernel) is put on a side stream beside the level-0 iterations of a
# 256-pair batch; the stage's wall time is printed for: no burn / burn before the iterations / burn beside them /
# beside them with the marching kernel's waves at raised priority / on a low-priority stream.
# Result of round 2 (profiles/r02c_overlap_probe.txt): beside == before == baseline + burn alone, in every variant.
set -e
cd "$(dirname "$0")/.."
cfg=${1:-2048,80000}
run() { timeout -k 10 120 python tools/bench_stage.py flow 257 3 2>&1 | grep -v amdgpu.ids; }
echo "== baseline"; run
echo "== burn $cfg before the iterations"; IPX_FI_OVERLAP_ALONE=1 IPX_FI_OVERLAP_PROBE=$cfg run
echo "== burn $cfg beside the iterations"; IPX_FI_OVERLAP_PROBE=$cfg run
echo "== beside, marching kernel at s_setprio 3"; IPX_FI_PRIO=1 IPX_FI_OVERLAP_PROBE=$cfg run
echo "== beside, side stream at the lowest priority"; IPX_FI_SIDE_LOW=1 IPX_FI_OVERLAP_PROBE=$cfg run
for k in 0 1; do echo "== f64 burn (kind $k: add / fma) before the iterations, 2048,20000: issue rate of f64 instructions"; IPX_FI_OVERLAP_F64=$k IPX_FI_OVERLAP_ALONE=1 IPX_FI_OVERLAP_PROBE=2048,20000 run; done

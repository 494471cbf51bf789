#!/usr/bin/env python3
"""python tools/kernel_us.py <bench.json>: value and the per-launch microseconds of the small kernels of a bench.py line."""
import json,sys
b=json.load(open(sys.argv[1])); k=b['kernels']
print(sys.argv[1], b['value'], ' '.join('%s=%.1f'%(n,k[n]['avg_us']) for n in ['ccl_init','ccl_merge','ccl_reduce','ccl_select','cluster_eval','close_cross3','finalize','static_maps'] if n in k))

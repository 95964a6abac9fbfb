#!/usr/bin/env python3
"""Average per-dispatch PMC counters of kernels matching a substring, from a rocprofv3 --pmc --output-format csv directory."""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'closed_loop')
agg = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            grid = r.get('Grid_Size_X') or r.get('Grid_Size') or '?'
            agg[(r['Kernel_Name'].split('(')[0][-60:] + ' grid=' + str(grid), r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    print(f'{k:76s} {c:24s} {sum(v) / len(v):16.0f}  n={len(v)}')

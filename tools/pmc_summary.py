#!/usr/bin/env python3
"""Average per-dispatch PMC counters of kernels matching a substring, from a rocprofv3 --pmc --output-format csv directory.
--halves <substring>: kernels whose name contains it are reported as two groups, the first and the second half of their dispatches in dispatch order
(bench.py launches the MCKF kernel for alpha = 1.5 first, then as often for alpha = 1.0: same name, same grid, different work)."""
import collections
import csv
import glob
import sys

halves = None
if '--halves' in sys.argv:
    i = sys.argv.index('--halves')
    halves = sys.argv[i + 1]
    del sys.argv[i:i + 2]
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'closed_loop')
rows = collections.defaultdict(list)
agg = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            grid = r.get('Grid_Size_X') or r.get('Grid_Size') or '?'
            key = (r['Kernel_Name'].split('(')[0][-100:] + ' grid=' + str(grid), r['Counter_Name'])
            if halves and halves in r['Kernel_Name']:
                rows[key].append((int(r.get('Dispatch_Id') or r.get('Dispatch_ID') or 0), float(r['Counter_Value'])))
            else:
                agg[key].append(float(r['Counter_Value']))
for (k, c), v in rows.items():
    v.sort()
    h = len(v) // 2
    name, grid = k.rsplit(' grid=', 1)
    agg[(name + ' [first half of the dispatches] grid=' + grid, c)] = [x for _, x in v[:h]]
    agg[(name + ' [second half of the dispatches] grid=' + grid, c)] = [x for _, x in v[h:]]
for (k, c), v in sorted(agg.items()):
    print(f'{k:76s} {c:24s} {sum(v) / len(v):16.0f}  n={len(v)}')

#!/usr/bin/env python3
"""Per-kernel (and per grid size: the same kernel serves BASELINE configs 2 and 3) duration statistics from a rocprofv3 --kernel-trace CSV, warm-up dispatches excluded.

usage: tools/trace_summary.py <dir with *kernel_trace.csv> [--skip name:count[:keep] ...]
For every kernel: calls, mean / median / min / max over ALL dispatches and over the dispatches that remain after dropping the first
`count` of the kernels named by --skip (substring match) -- bench.py's warm-up launches, which the --stats table averages in -- and,
with `:keep`, everything after the next `keep` dispatches (the timed launches; later launches of the same kernel belong to side objects)."""
import collections
import csv
import glob
import statistics
import sys

d = sys.argv[1]
skip = dict((a.split(':')[0], tuple(int(x) for x in a.split(':')[1:])) for a in sys.argv[3:]) if len(sys.argv) > 2 and sys.argv[2] == '--skip' else {}
rows = collections.defaultdict(list)
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        grid = r.get('Grid_Size_X') or r.get('Grid_Size') or '?'
        rows[r['Kernel_Name'] + ' grid=' + str(grid)].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
print('kernel,calls,mean_all_ms,skipped,mean_ms,median_ms,min_ms,max_ms')
for name, v in sorted(rows.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    if not name.startswith('void uvs::') and 'uvs::' not in name:
        continue
    v.sort()
    dur = [x[1] / 1e6 for x in v]
    plain = name.replace('void ', '').split('(')[0] + ' ' + name.split(' ')[-1]           # "uvs::kernel<...> grid=N": what the rules are matched against
    rule = max([c for k, c in skip.items() if k in plain] + [(0,)])
    n_skip = rule[0]
    kept = dur[n_skip:] if len(dur) > n_skip else dur
    if len(rule) > 1:
        kept = kept[:rule[1]]
    short = name.replace('void ', '').split('(')[0] + ' ' + name.split(' ')[-1]
    print(f'"{short}",{len(dur)},{statistics.mean(dur):.4f},{len(dur) - len(kept)},{statistics.mean(kept):.4f},{statistics.median(kept):.4f},{min(kept):.4f},{max(kept):.4f}')

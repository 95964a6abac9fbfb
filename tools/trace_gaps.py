#!/usr/bin/env python3
"""Busy / idle time of the GPU from a rocprofv3 --kernel-trace csv: kernels in start order, the gap before each, totals.
usage: tools/trace_gaps.py <dir with *_kernel_trace.csv> [last N dispatches]"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path))), key=lambda r: r[0])
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f'{len(rows)} dispatches, span {span / 1e6:.3f} ms, sum of kernel durations {busy / 1e6:.3f} ms')
prev = rows[0][0]
for s, e, name in rows:
    print(f'gap {max(s - prev, 0) / 1e3:8.1f} us  run {(e - s) / 1e3:9.1f} us  {name[:90]}')
    prev = max(prev, e)

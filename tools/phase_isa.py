#!/usr/bin/env python3
"""Instruction classes of the headline kernel's step loop, phase by phase: the -DUVS_STAMPS build pins an s_memtime between scheduling
fences at every phase boundary (rmckf_tuned.hpp, UVS_STAMP), so the assembly between two stamps is what that phase issues.
usage: hipcc ... -DUVS_QUICK -DUVS_STAMPS -S --cuda-device-only uvs_unity.hip -o /tmp/stamps.s ; tools/phase_isa.py /tmp/stamps.s"""
import collections
import re
import sys

KERNEL = '_ZN3uvs24closed_loop_tuned_kernelILi8ELi6ELi2ELi5ELi2ELi2ELb1ELb0ELb0EEEvNS_10ClosedArgsE'
# order of the stamps inside one trip of the loop (slot written at the END of the phase): loop top (5), plant (0), vmcnt probe (6), rows (1), control law (2), logs (3)
PHASES = ['loop edge: sincos advance, integrate q (stamp 5)', 'noise-load issue + plant (0)', 'vmcnt(0) probe (6)', 'row updates (1)', 'control law: QR + solve (2)',
          'logs + statistics (3)']


def classify(op):
    if op.startswith('v_accvgpr'): return 'accvgpr'
    if op.startswith('global_store'): return 'store'
    if op.startswith('global_load'): return 'load'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): return 'wait/nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_') and 'f64' in op:
        if 'fma' in op or 'fmac' in op: return 'f64 fma'
        if 'mul' in op: return 'f64 mul'
        if 'add' in op: return 'f64 add'
        return 'f64 other'
    if op.startswith('v_mov') and 'dpp' in op: return 'dpp mov'
    if op.startswith('v_'): return 'valu other'
    return 'other'


def main(path):
    s = open(path).read()
    a = s.index('\n' + KERNEL + ':')
    body = s[a:s.index('s_endpgm', a)].splitlines()
    labels = {l.strip()[:-1]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l.strip())}
    best = None                                                    # the step loop: the shortest backward branch whose span holds all six stamps
    for i, l in enumerate(body):
        m = re.match(r'\s*s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lo = labels[m.group(1)]
            if sum('s_memtime' in x for x in body[lo:i + 1]) == 6 and (best is None or i - lo < best[1] - best[0]):
                best = (lo, i)
    assert best is not None, 'no loop with six stamps found'
    loop = body[best[0]:best[1] + 1]
    cuts = [i for i, l in enumerate(loop) if 's_memtime' in l]
    # phase p = code between stamp p-1 and stamp p; the code after the last stamp wraps around to the first phase (loop edge)
    segs = [loop[cuts[-1] + 1:] + loop[:cuts[0]]] + [loop[cuts[i] + 1:cuts[i + 1]] for i in range(5)]
    cols = ['f64 fma', 'f64 mul', 'f64 add', 'f64 other', 'valu other', 'dpp mov', 'accvgpr', 'lds', 'salu', 'store', 'load', 'wait/nop']
    print(f'{"phase":58s}' + ''.join(f'{c:>11s}' for c in cols) + f'{"VALU":>8s}')
    tot = collections.Counter()
    for name, seg in zip(PHASES, segs):
        c = collections.Counter(classify(l.split()[0]) for l in seg if l.strip() and not l.strip().startswith(('.', ';', '_')) and not l.strip().endswith(':'))
        tot.update(c)
        valu = sum(c[k] for k in cols[:7])
        print(f'{name:58s}' + ''.join(f'{c[k]:11d}' for k in cols) + f'{valu:8d}')
    print(f'{"whole step (static count of the straight-line body)":58s}' + ''.join(f'{tot[k]:11d}' for k in cols) + f'{sum(tot[k] for k in cols[:7]):8d}')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/tmp/stamps.s')

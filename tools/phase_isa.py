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
    if op.startswith('v_permlane'): return 'permlane'
    if op.startswith('v_'): return 'valu other'
    return 'other'


WIDE_KERNEL = '_ZN3uvs23closed_loop_wide_kernelILi32ELi7ELi8ELi5ELb1ELb1ELi1EEEvNS_10ClosedArgsE'
# the wide kernel's stamps in loop order: loop edge (7), plant (0), rows (1), record stores (2), Gram + sums (3), Cholesky + refinement (5), logs (6)
WIDE_PHASES = ['loop edge: integrate q, clock (stamp 7)', 'noise issue + plant + measurement (0)', 'row updates + X into the record buffer (1)', 'record stores (2)',
               'Gram + rhs + 36 sums over 8 lanes (3)', 'Cholesky + solve + refinement + solve (5)', 'logs + statistics (6)']


def main(path):
    global KERNEL, PHASES
    if '--wide' in sys.argv:
        KERNEL, PHASES = WIDE_KERNEL, WIDE_PHASES
    NST = len(PHASES)
    s = open(path).read()
    a = s.index('\n' + KERNEL + ':')
    body = s[a:s.index('s_endpgm', a)].splitlines()
    labels = {re.match(r'^(\.LBB\d+_\d+):', l.strip()).group(1): i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l.strip())}
    # the step loop: the loop header label in front of a run of NST stamps; its last phase ends at the branch back to that label
    stamps = [i for i, l in enumerate(body) if 's_memtime' in l]
    best = None
    for lab, li in labels.items():
        inside = [i for i in stamps if i > li][:NST]
        if len(inside) < NST or any(labels[m] > li and labels[m] <= inside[-1] and 'Loop Header' in body[labels[m]] for m in labels if m != lab):
            continue
        back = [i for i, l in enumerate(body) if i > inside[-1] and re.match(r'\s*s_c?branch\w*\s+' + re.escape(lab) + r'\b', l)]
        if 'Loop Header' in body[li] and back and not [i for i in stamps if li < i < inside[0]]:
            if best is None or back[0] - li < best[1] - best[0]:
                best = (li, back[0], inside)
    assert best is not None, 'no loop with all the stamps found'
    lo, hi, cuts = best
    loop = body
    # phase p = code between stamp p-1 and stamp p; the code after the last stamp (to the back branch) plus the code from the header to the first stamp is the loop edge
    segs = [loop[cuts[-1] + 1:hi + 1] + loop[lo:cuts[0]]] + [loop[cuts[i] + 1:cuts[i + 1]] for i in range(NST - 1)]
    cols = ['f64 fma', 'f64 mul', 'f64 add', 'f64 other', 'valu other', 'dpp mov', 'permlane', 'accvgpr', 'lds', 'salu', 'store', 'load', 'wait/nop']
    print(f'{"phase":58s}' + ''.join(f'{c:>11s}' for c in cols) + f'{"VALU":>8s}')
    tot = collections.Counter()
    for name, seg in zip(PHASES, segs):
        c = collections.Counter(classify(l.split()[0]) for l in seg if l.strip() and not l.strip().startswith(('.', ';', '_')) and not l.strip().endswith(':'))
        tot.update(c)
        valu = sum(c[k] for k in cols[:8])
        print(f'{name:58s}' + ''.join(f'{c[k]:11d}' for k in cols) + f'{valu:8d}')
    print(f'{"whole step (static count of the straight-line body)":58s}' + ''.join(f'{tot[k]:11d}' for k in cols) + f'{sum(tot[k] for k in cols[:8]):8d}')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/tmp/stamps.s')

#!/usr/bin/env python3
"""Instruction-class histogram of kernels in the gfx950 assembly written by `make -C <pkg>/csrc asm`.
usage: tools/isa_stats.py [substring-of-mangled-name ...]"""
import collections
import os
import re
import sys

ASM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'uncalibrated-visual-servoing_amd', 'csrc', 'uvs_rmckf.gfx950.s')


def classify(op):
    if op.startswith('v_accvgpr'): return 'accvgpr mov'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('global_load'): return 'global_load'
    if op.startswith('global_store'): return 'global_store'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane<->sgpr'
    if op.startswith('v_mov') or op.startswith('v_dual_mov'): return 'v_mov'
    if op.startswith('v_cndmask'): return 'v_cndmask'
    if 'f64' in op:
        if op.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64')): return 'f64 fma/mul/add'
        if op.startswith(('v_div', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_trig', 'v_ldexp', 'v_frexp', 'v_rndne', 'v_cvt', 'v_fract', 'v_floor')): return 'f64 special'
        return 'f64 other'
    if op.startswith('v_'): return 'valu int/other'
    return 'other'


def main():
    text = open(ASM).read()
    want = sys.argv[1:] or ['closed_loop']
    for fn in re.split(r'\n(?=_ZN3uvs\w+:\s)', text):
        name = fn.split(':', 1)[0]
        if not name.startswith('_ZN3uvs') or not any(w in name for w in want):
            continue
        ops = []
        for line in fn.split('\n'):
            line = line.strip()
            if not line or line[0] in '.;_' or line.endswith(':'):
                continue
            ops.append(line.split()[0])
        hist = collections.Counter(classify(o) for o in ops)
        print(f'{name}: {len(ops)} instructions')
        for k, v in hist.most_common():
            print(f'    {k:18s} {v:6d}  {100.0 * v / len(ops):5.1f}%')


if __name__ == '__main__':
    main()

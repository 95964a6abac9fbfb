#!/usr/bin/env python3
"""Instruction-class histogram of kernels in the gfx950 assembly written by `make -C <pkg>/csrc asm`.

usage: tools/isa_stats.py [--loop] [--ops] substring-of-mangled-name ...
  --loop  restrict to the largest loop (label .. backward branch) of each kernel: the per-step body
  --ops   also print the 40 most frequent opcodes
"""
import collections
import os
import re
import sys

ASM = os.environ.get("UVS_ASM", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uncalibrated-visual-servoing_amd", "csrc", "uvs_rmckf.gfx950.s"))


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
    if op.startswith('v_mov') and 'dpp' in op: return 'v_mov dpp'
    if op.startswith('v_mov'): return 'v_mov'
    if op.startswith('v_cndmask'): return 'v_cndmask'
    if 'f64' in op:
        if op.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64')): return 'f64 fma/mul/add'
        if op.startswith(('v_div', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_trig', 'v_ldexp', 'v_frexp', 'v_rndne', 'v_cvt', 'v_fract', 'v_floor')): return 'f64 special'
        return 'f64 other'
    if op.startswith('v_'): return 'valu int/other'
    return 'other'


def instructions(fn):
    """[(label-or-None, opcode, operands)] in program order."""
    out = []
    for line in fn.split('\n'):
        line = line.split(';')[0].strip()
        if not line or line[0] == '.' and not line.endswith(':'):
            continue
        if line.endswith(':'):
            out.append((line[:-1], None, None))
            continue
        parts = line.split(None, 1)
        op = parts[0]
        if 'dpp' in line and op.startswith('v_mov'):
            op += '_dpp'
        out.append((None, op, parts[1] if len(parts) > 1 else ''))
    return out


def largest_loop(ins):
    labels = {lab: i for i, (lab, op, _) in enumerate(ins) if lab}
    best = (0, 0)
    for i, (lab, op, args) in enumerate(ins):
        if op and op.startswith('s_cbranch') or op == 's_branch':
            tgt = args.strip()
            if tgt in labels and labels[tgt] < i and i - labels[tgt] > best[1] - best[0]:
                best = (labels[tgt], i)
    return best


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    loop, show_ops = '--loop' in sys.argv, '--ops' in sys.argv
    text = open(ASM).read()
    want = args or ['closed_loop']
    for fn in re.split(r'\n(?=_ZN3uvs\w+:\s)', text):
        name = fn.split(':', 1)[0]
        if not name.startswith('_ZN3uvs') or not any(w in name for w in want):
            continue
        ins = instructions(fn)
        if loop:
            lo, hi = largest_loop(ins)
            ins = ins[lo:hi + 1]
        ops = [op for lab, op, _ in ins if op]
        hist = collections.Counter(classify(o) for o in ops)
        print(f'{name}: {len(ops)} instructions' + (' in the largest loop' if loop else ''))
        for k, v in hist.most_common():
            print(f'    {k:18s} {v:6d}  {100.0 * v / len(ops):5.1f}%')
        if show_ops:
            for k, v in collections.Counter(ops).most_common(40):
                print(f'        {k:28s} {v:6d}')


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""profiles/traffic_latest.json from a round's pmc_summary.txt + kernel_trace_summary.csv (tools/profile_round.sh): HBM bytes per launch
(FETCH_SIZE KiB x 1024 x 2 per the gfx950 correction + WRITE_SIZE KiB x 1024), VALU wave-instructions and fp64 FLOP per launch of the
headline kernel, and the same for configs 3 / 5 and the replay kernels.  usage: tools/make_traffic_json.py profiles/r03 3"""
import csv
import json
import os
import re
import sys

d, rnd = sys.argv[1], int(sys.argv[2])
cnt = {}
for line in open(os.path.join(d, 'pmc_summary.txt')):
    m = re.match(r'(.*?) grid=(\d+)\s+(\S+)\s+(\d+)\s+n=(\d+)', line)
    if m:
        cnt[(m.group(1).strip(), int(m.group(2)), m.group(3))] = int(m.group(4))
ms = {}
for r in csv.DictReader(open(os.path.join(d, 'kernel_trace_summary.csv'))):
    name, grid = r['kernel'].rsplit(' grid=', 1)
    ms[(name.strip(), int(grid))] = float(r['mean_ms'])


def find(table, sub, grid):
    hits = [k for k in table if sub in k[0] and k[1] == grid]
    return hits


def same_kernel(sub, name):
    """`sub` (a piece of a kernel name, e.g. 'tuned_kernel<8, 6, 2, 5, 2, 2, true') against a name of pmc_summary.txt, which keeps only the tail of long
    names: compare from 'kernel<' on, and the letter before '_kernel<' (closed_loop_tune-d / wid-e) where the summary still has it."""
    if sub in name:
        return True
    if 'kernel<' not in sub or 'kernel<' not in name:
        return False
    st, nt = sub[sub.index('kernel<'):], name[name.index('kernel<'):]
    if not nt.startswith(st):
        return False
    sh, nh = sub[:sub.index('kernel<')].rstrip('_'), name[:name.index('kernel<')].rstrip('_')
    return not nh or not sh or sh[-1] == nh[-1]


def entry(sub, grid, alg_bytes, label):
    def c(name):
        ks = [k for k in cnt if same_kernel(sub, k[0]) and k[1] == grid and k[2] == name]
        return cnt[ks[0]] if ks else None
    fetch, write = c('FETCH_SIZE'), c('WRITE_SIZE')
    out = {'kernel': label, 'fetch_size_kib': fetch, 'write_size_kib': write}
    if fetch is not None and write is not None:
        out.update(fetch_bytes_corrected=fetch * 1024 * 2, write_bytes=write * 1024, hbm_bytes_per_launch=fetch * 1024 * 2 + write * 1024)
    out['algorithmic_bytes_per_launch'] = alg_bytes
    valu = c('SQ_INSTS_VALU')
    if valu:
        out['valu_wave_instr_per_launch'] = valu
    cls = {k: c('SQ_INSTS_VALU_' + k + '_F64') for k in ('FMA', 'MUL', 'ADD', 'TRANS')}
    if all(v is not None for v in cls.values()):
        out['fp64_wave_instr_per_launch'] = cls
        out['fp64_flop_per_launch'] = 64 * (2 * cls['FMA'] + cls['MUL'] + cls['ADD'] + cls['TRANS'])
    for extra in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_WR'):
        if c(extra) is not None:
            out[extra] = c(extra)
    t = [v for k, v in ms.items() if sub in k[0] and k[1] == grid]
    if t:
        out['avg_kernel_ms_rocprof'] = t[0]
    return out


U2, U3, K = 65536 * 299, 262144 * 299, 299
head = entry('tuned_kernel<8, 6, 2, 5, 2, 2, true, false, false, false, false, false>', 131072, U2 * 560, 'closed_loop_tuned_kernel<8,6,2,GMCKF,DH(axis-aligned),2,true>')
doc = {'round': rnd, 'workload': 'BASELINE config 2, 65536 trials x 299 updates, X+err+q logged, layout kct', 'source': d + '/pmc_summary.txt',
       'note': 'rocprofv3 --pmc passes with --kernel-trace only (tools/profile_round.sh), per-dispatch averages; FETCH_SIZE doubled per the gfx950 correction '
               '(MI355X_MICROARCH.md, HBM section: documented for 16 B/lane reads; these are 8 B/lane and the doubled figure lands on the algorithmic read bytes '
               'within 0.5 %, which supports applying it here); fp64 FLOP = 64 lanes x (2 FMA + MUL + ADD + TRANS) wave-instructions'}
doc.update(head)
ver = os.path.join(d, 'library_version.txt')                     # written on the GPU box by tools/profile_round.sh: the library the counters were taken on
doc['library_version'] = open(ver).read().strip() if os.path.exists(ver) else None
doc['config3'] = entry('tuned_kernel<8, 6, 2, 5, 2, 2, true, false, false, false, false, false>', 524288, U3 * 560, 'closed_loop_tuned_kernel<8,6,2,GMCKF,DH(axis-aligned),2,true>')
doc['config5'] = entry('closed_loop_wide_kernel<32, 7, 8, 5, true, true', 524288, U2 * 2360, 'closed_loop_wide_kernel<32,7,8,GMCKF,true,true>')
# (MCKF since round 4: 8 work items per trial chunk, grid 2048 x 8 x 64 -- its hand-over traffic, 141 doubles per lane and segment edge out and back, is part of the counters)
# KF / IMCC-KF (round 6): a trial-fastest X stream leaves through the pair-store instantiation (XREC + XPAIR)
doc['other_estimators'] = {name: entry(f'tuned_kernel<8, 6, 2, {code}, 2, 2, true, false, false, true, false, true>', 131072, U2 * 560,
                                       f'closed_loop_tuned_kernel<8,6,2,{name},DH(axis-aligned),2,true,XREC,XPAIR> (16-byte pair stores)')
                           for name, code in (('KF', 2), ('IMCCKF', 4))}
# MCKF: bench.py launches the same kernel for alpha = 1.5 and then as often for alpha = 1.0; tools/pmc_summary.py --halves keeps the two apart.  The kernel-trace
# mean mixes them and is dropped here (the bench line holds both times).
for key, tag, alpha in (('MCKF', '[first half', 1.5), ('MCKF_alpha1', '[second half', 1.0)):
    hits = [k for k in cnt if same_kernel('tuned_kernel<8, 6, 2, 3, 2, 2, true', k[0]) and tag in k[0] and k[1] == 1048576 and k[2] == 'SQ_INSTS_VALU']
    if hits:
        e = entry(hits[0][0], 1048576, U2 * 560, f'closed_loop_tuned_kernel<8,6,2,MCKF,DH(axis-aligned),2,true> (8 segments per trial), alpha = {alpha}')
        e.pop('avg_kernel_ms_rocprof', None)
        doc['other_estimators'][key] = e
doc['replay'] = {'estimator_only': entry('replay_rows_kernel<8, 6, 4, 5, true, true, true, false, 0>', 262144, U2 * 560, 'replay_rows_kernel<8,6,4,GMCKF,true,true,BYWAVE>'),
                 'estimator_only_records': entry('replay_rows_kernel<8, 6, 4, 5, true, true, false, true, 0>', 262144, U2 * 560, 'replay_rows_kernel<8,6,4,GMCKF,true,true,false,REC>'),
                 'estimator_only_f32': entry('replay_f32_kernel<5, true, true>', 131072, U2 * 280, 'replay_f32_kernel<GMCKF,true,true> (fp32 measured variant)'),
                 'estimator_and_control_law': entry('replay_rows_kernel<8, 6, 4, 5, true, true, true, false, 2>', 393216, U2 * 608, 'replay_rows_kernel<8,6,4,GMCKF,true,true,BYWAVE,false,CW=2>')}
json.dump(doc, open(os.path.join(os.path.dirname(d.rstrip('/')), 'traffic_latest.json'), 'w'), indent=1)
print(json.dumps({k: v for k, v in doc.items() if not isinstance(v, dict)}, indent=1))

#!/usr/bin/env python3
"""The reference's headline experiment (results/results1.fig: ITAE against alpha for KF, MCKF, IMCC-KF and RMCKF; main.py:104-196 + results/plot_errorbar.m) run
on the GPU at Monte-Carlo sizes the reference cannot reach: `batch.run_sweep` of the reference's config (fixed start, results1 protocol) with `epoch` trials per
cell, per-cell success count and mean / std / median of ||ITAE|| as plot_errorbar.m reduces them -- next to the reference's own 100-trial table
(tests/golden/sweep_r1_*.npz, produced by the unmodified main.py).  usage (GPU box): python tools/results_table.py [epoch] [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import uvs_amd as uvs  # noqa: E402

epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
out_path = sys.argv[2] if len(sys.argv) > 2 else None
table = {'epoch': epoch, 'cells': None, 'estimators': {}}
for name, label in (('KF', 'KF'), ('MCKF', 'MCKF'), ('IMCCKF', 'IMCC-KF'), ('GMCKF', 'RMCKF')):
    z = np.load(os.path.join(ROOT, 'tests', 'golden', f'sweep_r1_{name.lower()}.npz'))
    cfg = json.loads(str(z['config']))
    cfg.pop('_provenance', None)
    cfg['experiments']['epoch'] = epoch
    uvs.batch.run_sweep(cfg, epoch=min(epoch, 4096))                 # warm-up
    t0 = time.perf_counter()
    res = uvs.batch.run_sweep(cfg)
    wall = time.perf_counter() - t0
    summ = res.cell_summary()
    table['cells'] = [float(c) for c in res.plan.cells]
    table['estimators'][label] = {
        'gpu_seconds_for_the_sweep': wall, 'trials': len(res.plan), 'updates': int(res.k_done.sum()),
        'gpu': [{'success': summ[c]['success'], 'itae_mean': summ[c]['itae_mean'], 'itae_std': summ[c]['itae_std'], 'itae_median': summ[c]['itae_median']} for c in range(12)],
        'reference_100_trials': [{'success': int(z['cell_n_success'][c]), 'itae_mean': float(z['cell_mean'][c, 2]), 'itae_std': float(z['cell_std'][c, 2]),
                                  'itae_median': float(z['cell_median'][c, 2])} for c in range(12)],
        'reference_seconds_for_100_trials_per_cell': json.loads(str(z['config']))['_provenance']['reference_seconds']}
    print(f'{label}: {len(res.plan)} trials ({epoch} per cell) in {wall:.2f} s on the GPU; the reference took {table["estimators"][label]["reference_seconds_for_100_trials_per_cell"]:.0f} s for 100 per cell')
    print(f'  {"alpha":>6s} {"FAIL %":>8s} {"median ITAE":>12s} {"mean ITAE":>12s}   | reference, 100 trials: {"FAIL":>5s} {"median":>10s} {"mean":>10s}')
    for c in range(12):
        g, r = table['estimators'][label]['gpu'][c], table['estimators'][label]['reference_100_trials'][c]
        print(f'  {table["cells"][c]:6.3f} {100 * (1 - g["success"] / epoch):8.2f} {g["itae_median"]:12.0f} {g["itae_mean"]:12.0f}   |{"":25s}{100 - r["success"]:5d} {r["itae_median"]:10.0f} {r["itae_mean"]:10.0f}')
if out_path:
    json.dump(table, open(out_path, 'w'), indent=1)

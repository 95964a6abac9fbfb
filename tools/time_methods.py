#!/usr/bin/env python3
"""Closed-loop sweep of BASELINE config 2 (65 536 trials x 299 updates, X + err + q logged) for each of the reference's estimators:
average kernel time over a few launches.  Run on the GPU box; UVS_LIB_PATH selects an experiment build."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import uvs_amd
from uvs_amd import engine, batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--methods', default='GMCKF,KF,IMCCKF,MCKF')
ap.add_argument('--trials', type=int, default=65536)
ap.add_argument('--reps', type=int, default=8)
ap.add_argument('--lanes', type=int, default=0)
ap.add_argument('--alpha', type=float, default=1.5)
ap.add_argument('--x-layout', default=None, help="layout of the X stream alone ('ktc' = per-trial records)")
ap.add_argument('--segments', type=int, default=0, help='segments per trial (bits 8-15 of fp.reserved): 0 = library choice, 1 = whole trials')
ap.add_argument('--reserved', type=int, default=0, help='extra bits OR-ed into fp.reserved (experiment switches of diagnostic builds)')
ap.add_argument('--strict', action='store_true', help='UVS_OPT_STRICT_PINV: every solve certified / through the careful kernels')
args = ap.parse_args()
T, dev = args.trials, torch.device('cuda')
cfg = bench.config2()
cfg['experiments']['epoch'] = T
K = len(engine.loop_clock(0.05, 15))
cfg['noise']['noise_params']['alpha'] = args.alpha
plan = batch.plan_trials(cfg, cells=[args.alpha])
noise = batch.device_noise(cfg, plan, 0, T, K, dev, share=False)
q0 = torch.as_tensor(plan.q_start.copy(), device=dev)
plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
if args.reps > 1000:
    open('gpurun_out/.probe_started', 'w').write('1')                # tools/power_probe.sh waits for this
for meth in args.methods.split(','):
    fp = engine.make_params(8, 6, meth, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, args.lanes)
    fp.reserved = (args.segments << 8) | (1 if args.strict else 0) | args.reserved
    ms = []
    for i in range(2 + args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'), x_layout=args.x_layout)
        e1.record(); torch.cuda.synchronize()
        if i >= 2:
            ms.append(e0.elapsed_time(e1))
    upd = int(out['k_done'].sum())
    print(f'{meth:7s} lanes={args.lanes} x_layout={args.x_layout}: {np.mean(ms):.3f} ms (min {np.min(ms):.3f})  {upd / np.mean(ms) / 1e6:.2f} G updates/s  {upd * 560 / np.mean(ms) / 1e9:.2f} TB/s  failed {int((out["status"] != 0).sum())}')

#!/usr/bin/env python3
"""BASELINE config 5 (16-feature / 7-DoF stress shape, 65 536 trials x 299 updates, X + err + q logged, per-trial records): kernel time per launch.
usage (GPU box): [UVS_LIB_PATH=...] python tools/time_config5.py [--reps 5] [--method GMCKF]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import uvs_amd

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--method', default='GMCKF')
ap.add_argument('--trials', type=int, default=65536)
args = ap.parse_args()
T, K, M, N = args.trials, 299, 32, 7
lin = uvs_amd.LinearPlant.random(M, N, seed=2)
rng = np.random.default_rng(5)
q_goal = lin.q0 + rng.uniform(-0.3, 0.3, N)
q0 = torch.as_tensor(q_goal + np.random.default_rng(12345).uniform(-0.15, 0.15, (T, N)), device='cuda')
x0 = torch.as_tensor(np.tile((lin.J * (1 + 0.1 * rng.normal(size=lin.J.shape))).ravel(), (T, 1)), device='cuda')
noise = uvs_amd.noise_device.generate(uvs_amd.NoiseType.ALPHA_STABLE, dict(alpha=1.5, beta=0, gamma=1, delta=0), 123456 + np.arange(T), M, K, layout='ktc', device='cuda')
plant = lin.to_struct('cuda')
for meth in args.method.split(','):
    fp = uvs_amd.engine.make_params(M, N, meth, 10, False, 0.05, 15, 0.2, lin.features(q_goal), False, 0)
    ms = []
    for i in range(2 + args.reps):
        out = uvs_amd.engine.closed_loop(fp, plant, q0, noise, x0, want=('x', 'err', 'q'), layout='ktc')
        torch.cuda.synchronize()
        if i >= 2:
            ms.append(out['events'][0].elapsed_time(out['events'][1]))
    upd = int(out['k_done'].sum())
    print(f'config 5 {meth:7s}: {np.mean(ms):.3f} ms (min {np.min(ms):.3f})  {upd * 2360 / np.mean(ms) / 1e9 / 8:.4f} of 8 TB/s  failed {int((out["status"] != 0).sum())}', flush=True)
    del out

#!/usr/bin/env python3
"""How large are the joint steps |dq dt| of a closed-loop run, and how many wavefront-steps see one above the incremental sincos' bound
(rmckf_math.hpp: kSinCosStepMax = 0.1 rad -> the wavefront re-seeds its pairs with the full sincos at the next step)?
usage (GPU box): python tools/step_histogram.py [--config 2|3] [--hold] [--trials T]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import uvs_amd
from uvs_amd import engine, batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--config', type=int, default=3)
ap.add_argument('--hold', action='store_true')
ap.add_argument('--trials', type=int, default=65536)
args = ap.parse_args()
T, dev = args.trials, torch.device('cuda')
cfg = bench.config2()
cfg['experiments']['epoch'] = T
anneal = False
if args.config == 3:
    cfg['noise'].update(type='GAUSSIAN_MIXTURE', noise_params={'std': 1.0, 'mean': 50.0, 'rho': 0.1}, hold=bool(args.hold), hold_time=0.5)
    anneal = True
K = len(engine.loop_clock(0.05, 15))
cells = [0.1] if args.config == 3 else [1.5]
plan = batch.plan_trials(cfg, cells=cells)
noise = batch.device_noise(cfg, plan, 0, T, K, dev)
q0 = torch.as_tensor(plan.q_start.copy(), device=dev)
plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
fp = engine.make_params(8, 6, 'GMCKF', 10, anneal, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0)
out = engine.closed_loop(fp, plant, q0, noise, want=('q',))
q = engine.as_tkc(out['q'], 'kct')                                # [trial][step][joint]
step = (q[:, 1:, :] - q[:, :-1, :]).abs().amax(dim=2)             # largest joint step of a trial at a step
step = torch.nan_to_num(step, nan=0.0, posinf=1e9)
wf = step.reshape(T // 32, 32, K - 1).amax(dim=1)                 # per wavefront (32 trials) and step
print(f'config {args.config} hold={args.hold}: {T} trials, failed {int((out["status"] != 0).sum())}')
for thr in (0.05, 0.1, 0.2, 0.5, 1.0, 3.0):
    print(f'  |dq dt| > {thr:4.2f} rad: {float((step > thr).float().mean()) * 100:8.4f} % of trial-steps, {float((wf > thr).float().mean()) * 100:8.3f} % of wavefront-steps')
big = (step > 0.1).any(dim=1)
print(f'  trials with at least one step > 0.1 rad: {int(big.sum())} ({float(big.float().mean()) * 100:.2f} %); steps above per such trial: {float((step[big] > 0.1).float().sum(dim=1).mean()) if big.any() else 0:.1f}')

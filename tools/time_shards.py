#!/usr/bin/env python3
"""Kernel time of ONE GPU on the shard a rank of north_star's strong series would own (65 536 trials / N), per lanes-per-filter variant:
what the series starts from on each rank -- one-GPU shard timings, NOT a scaling curve (no gather, no second GPU).
usage (GPU box): [UVS_LIB_PATH=...] python tools/time_shards.py [--lanes 2,4] [--trials 8192,16384,32768,65536]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import uvs_amd
from uvs_amd import engine, batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--lanes', default='2,4')
ap.add_argument('--trials', default='8192,16384,32768,65536')
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--method', default='GMCKF')
ap.add_argument('--latency', action='store_true', help='UVS_OPT_LATENCY (with --lanes 0)')
args = ap.parse_args()
dev = torch.device('cuda')
K = len(engine.loop_clock(0.05, 15))
for T in [int(t) for t in args.trials.split(',')]:
    cfg = bench.config2()
    cfg['experiments']['epoch'] = T
    plan = batch.plan_trials(cfg, cells=[1.5])
    noise = batch.device_noise(cfg, plan, 0, T, K, dev)
    q0 = torch.as_tensor(plan.q_start.copy(), device=dev)
    plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    for L in [int(x) for x in args.lanes.split(',')]:
        fp = engine.make_params(8, 6, args.method, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, L)
        if args.latency:
            fp.reserved = 2
        ms = []
        for i in range(3 + args.reps):
            out = engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
            torch.cuda.synchronize()
            if i >= 3:
                ms.append(out['events'][0].elapsed_time(out['events'][1]))
        import ctypes as C
        used = int(uvs_amd.lib().uvs_rmckf_closed_loop_lanes(C.byref(fp), C.byref(plant), T))
        print(f'trials {T:6d} lanes {L} (kernel: {used} per filter{", two-lane bits" if L == 0 and used == 4 else ""}): {np.mean(ms):.3f} ms (min {np.min(ms):.3f}); '
              f'wavefronts {T * used // 64}; failed {int((out["status"] != 0).sum())}', flush=True)
    del noise

#!/usr/bin/env python3
"""MCKF closed loop (BASELINE config-2 workload) cut into n segments per trial (uvs_rmckf_closed_loop_ws_f64, bits 8-15 of fp.reserved):
kernel time per launch for each n, and a bit-for-bit comparison of every output with the unsegmented launch.
usage (GPU box): python tools/time_mckf_segments.py [--alpha 1.5,1.0] [--trials 65536] [--segments 1,2,3,4]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import uvs_amd
from uvs_amd import engine, batch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--alpha', default='1.5,1.0')
ap.add_argument('--trials', default='65536')
ap.add_argument('--segments', default='1,2,3,4')
ap.add_argument('--reps', type=int, default=6)
ap.add_argument('--method', default='MCKF')
args = ap.parse_args()
dev = torch.device('cuda')
K = len(engine.loop_clock(0.05, 15))
for T in [int(t) for t in args.trials.split(',')]:
    for alpha in [float(a) for a in args.alpha.split(',')]:
        cfg = bench.config2()
        cfg['experiments']['epoch'] = T
        cfg['noise']['noise_params']['alpha'] = alpha
        plan = batch.plan_trials(cfg, cells=[alpha])
        noise = batch.device_noise(cfg, plan, 0, T, K, dev)
        q0 = torch.as_tensor(plan.q_start.copy(), device=dev)
        plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
        ref = None
        for n in [int(x) for x in args.segments.split(',')]:
            fp = engine.make_params(8, 6, args.method, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0)
            fp.reserved = n << 8
            ms = []
            for i in range(2 + args.reps):
                out = engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
                torch.cuda.synchronize()
                if i >= 2:
                    ms.append(out['events'][0].elapsed_time(out['events'][1]))
            kd = out['k_done']
            live = torch.arange(K, device=dev)[:, None, None] < kd[None, None, :]              # rows at and after k_done are unspecified
            cur = {k: torch.where(live, out[k], torch.zeros_like(out[k])) for k in ('x', 'err', 'q')}
            cur.update(stats=out['stats'].clone(), status=out['status'].clone(), k_done=kd.clone())
            same = 'reference'
            if ref is None:
                ref = cur
            else:
                same = 'bit-identical' if all(torch.equal(ref[k].view(torch.int64) if ref[k].dtype == torch.float64 else ref[k],
                                                          cur[k].view(torch.int64) if cur[k].dtype == torch.float64 else cur[k]) for k in ref) else 'DIFFERENT'
            upd = int(kd.sum())
            print(f'T {T} alpha {alpha} segments {n}: {np.mean(ms):.3f} ms (min {np.min(ms):.3f}) {upd * 560 / np.mean(ms) / 1e9 / 8:.3f} of 8 TB/s, failed {int((out["status"] != 0).sum())}, {same}', flush=True)
        del noise, ref, cur, out
        torch.cuda.empty_cache()

"""Analysis helper (test infrastructure, runs the C oracle): how often does a 32-trial wavefront of the MCKF kernel walk the
fixed-point branch, with how many filters at once and for how many passes?  `python tests/fpi_branch_stats.py [alpha] [trials]`."""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import uvs_amd
from uvs_amd import batch, engine
from oracle import c_oracle

alpha = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = bench.config2()
cfg['experiments']['epoch'] = T
cfg['estimator']['method'] = 'MCKF'
cfg['noise']['noise_params']['alpha'] = alpha
plan = batch.plan_trials(cfg, cells=[alpha])
K = len(engine.loop_clock(0.05, 15))
noise = np.empty((T, K, 8))
batch.trial_noise(cfg, plan, 0, T, K, noise)
out = c_oracle.closed_loop_batch(plan.q_start, noise, cfg['experiments']['desired_f'], 'MCKF')
fpi, kd = out['fpi'], out['k_done']
live = np.arange(K)[None, :] < kd[:, None]
extra = np.where(live, np.maximum(fpi - 1, 0), 0)             # passes beyond the first
print(f'alpha {alpha}, {T} trials: failed {int((out["status"] != 0).sum())}, trial-steps live {int(live.sum())}, iterating {(extra > 0).sum()} '
      f'({(extra > 0).sum() / live.sum():.4%}), extra passes hist {np.bincount(extra[extra > 0])[:12]} max {extra.max()}')
w = extra[:T // 32 * 32].reshape(-1, 32, K)
it_w = (w > 0).sum(1)                                         # filters iterating per wavefront-step
fire = it_w > 0
print(f'wavefront-steps {fire.size}, firing {fire.sum()} ({fire.mean():.3%}); filters per firing hist {np.bincount(it_w[fire])}; '
      f'passes of a firing (max over its filters) mean {w.max(1)[fire].mean():.2f}; sum over its filters mean {w.sum(1)[fire].mean():.2f}')

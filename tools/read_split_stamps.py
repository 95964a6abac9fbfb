#!/usr/bin/env python3
"""Run BASELINE config 2 once on the -DUVS_SPLIT_STAMPS build and print where the wavefronts of the role-split kernel spend a step.
usage (GPU box): UVS_LIB_PATH=<pkg>/libuvs_split_stamps.so python tools/read_split_stamps.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import uvs_amd  # noqa: E402
import bench  # noqa: E402

T, K = 65536, 299
cfg = bench.config2()
plan = uvs_amd.batch.plan_trials(cfg, cells=[1.5])
fp = uvs_amd.engine.make_params(8, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 5)
noise = torch.empty((K, 8, T), dtype=torch.float64, device='cuda').normal_()
q0 = torch.as_tensor(plan.q_start, device='cuda')
plant = uvs_amd.SyntheticPlant.ur10().to_struct()
for _ in range(2):
    out = uvs_amd.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
torch.cuda.synchronize()
stats = out['stats'].cpu().numpy().ravel()
groups = T // 64
st = np.stack([stats[3 * 64 * g: 3 * 64 * g + 32].reshape(4, 8) for g in range(groups)])          # [group][wave][slot]
ids = np.stack([stats[3 * 64 * g + 32: 3 * 64 * g + 36] for g in range(groups)]).astype(np.int64)
names = ['estimator phase', 'wait at barrier 1', 'control turn', 'logging turn', 'books', 'wait at barrier 2']
tot = st[:, :, 7].mean()
rt = st[:, :, 6].mean()
print(f'groups {groups}; cycles per wavefront {tot:.0f} = {tot / K:.0f} per step; loop wall {rt / 100e6 * 1e3:.3f} ms per wavefront; shader clock {tot / (rt / 100e6) / 1e9:.3f} GHz')
for i, n in enumerate(names):
    v = st[:, :, i].mean() / K
    print(f'  {n:20s} {v:8.0f} cycles/step (average over the 4 wavefronts)  {100 * v / (tot / K):5.1f}%')
print(f'  control turn, when it is this wavefront\'s: {4 * st[:, :, 2].mean() / K:.0f} cycles;  logging turn, when logging: {4 / 3 * st[:, :, 3].mean() / K:.0f} cycles')
sub = np.stack([stats[3 * 64 * g + 40: 3 * 64 * g + 60].reshape(4, 5) for g in range(groups)])
for i, n in enumerate(['park block in LDS', 'normal equations + refinement', 'q update + q/dq logs', 'kinematics + projection', 'unpark']):
    print(f'    control turn / {n:32s} {4 * sub[:, :, i].mean() / K:8.0f} cycles')
simd = (ids >> 4) & 3
cu = (ids >> 8) & 15
print('SIMD of waves 0..3, first groups:', simd[:6].tolist(), ' distinct SIMDs per group: ', np.bincount([len(set(r)) for r in simd.tolist()], minlength=5).tolist())

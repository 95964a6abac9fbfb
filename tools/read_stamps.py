#!/usr/bin/env python3
"""Run BASELINE config 2 once on the -DUVS_STAMPS build and print the per-phase cycle shares of the tuned kernel; --wide: config 5 on the wide kernel.
usage (GPU box): UVS_LIB_PATH=tools/diag/libuvs_stamps.so python tools/read_stamps.py [lanes | --wide]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import uvs_amd  # noqa: E402
import bench  # noqa: E402

if '--wide' in sys.argv:                                          # config 5: (32,7), 8 lanes per filter, 8 trials per wavefront, 24 stamp doubles per wavefront
    T, K, M, N = 65536, 299, 32, 7
    lin = uvs_amd.LinearPlant.random(M, N, seed=2)
    rng = np.random.default_rng(5)
    q_goal = lin.q0 + rng.uniform(-0.3, 0.3, N)
    fp = uvs_amd.engine.make_params(M, N, 'GMCKF', 10, False, 0.05, 15, 0.2, lin.features(q_goal), False, 0)
    q0 = torch.as_tensor(q_goal + np.random.default_rng(12345).uniform(-0.15, 0.15, (T, N)), device='cuda')
    x0 = torch.as_tensor(np.tile((lin.J * (1 + 0.1 * rng.normal(size=lin.J.shape))).ravel(), (T, 1)), device='cuda')
    noise = torch.empty((K, T, M), dtype=torch.float64, device='cuda').normal_()
    for _ in range(2):
        out = uvs_amd.engine.closed_loop(fp, lin.to_struct('cuda'), q0, noise, x0, want=('x', 'err', 'q'), layout='ktc')
    torch.cuda.synchronize()
    stats = out['stats'].cpu().numpy().ravel()
    waves = T // 8
    st = np.stack([stats[24 * w: 24 * w + 8] for w in range(waves)])
    names = ['noise issue + plant + measurement', 'row updates (+ X into the record buffer)', 'record stores', 'Gram + rhs + 36 sums', '(100 MHz wall ticks)',
             'Cholesky + solve + refinement', 'logs + statistics', 'loop edge']
    rt = st[:, 4].copy()
    st[:, 4] = 0
    tot = st.sum(axis=1)
    print(f'waves {waves}, cycles per wave: mean {tot.mean():.0f}  min {tot.min():.0f}  max {tot.max():.0f};  per step {tot.mean() / K:.0f}')
    print(f'shader clock while the kernel runs: {tot.mean() / (rt.mean() / 100e6) / 1e9:.3f} GHz  (loop wall time {rt.mean() / 100e6 * 1e3:.3f} ms per wavefront)')
    for i, n in enumerate(names):
        if i != 4:
            print(f'  {n:44s} {st[:, i].mean() / K:9.0f} cycles/step  {100 * st[:, i].mean() / tot.mean():5.1f}%')
    sys.exit(0)
if '--items' in sys.argv:                                          # work items of a segmented MCKF launch (-DUVS_ITEM_STAMPS build): entry -> ready -> steps -> handed over
    alpha = 1.5
    T, K = 65536, 299
    cfg = bench.config2()
    plan = uvs_amd.batch.plan_trials(cfg, cells=[alpha])
    noise = uvs_amd.batch.device_noise(cfg, plan, 0, T, K, 'cuda', share=False)
    q0 = torch.as_tensor(plan.q_start, device='cuda')
    plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    for nseg in (1, 8):
        fp = uvs_amd.engine.make_params(8, 6, 'MCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 2)
        fp.reserved = nseg << 8
        out = uvs_amd.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))          # warm-up (the next call starts from zeroed stats)
        out = uvs_amd.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
        torch.cuda.synchronize()
        st = out['stats'].cpu().numpy().ravel()[:8 * nseg].reshape(nseg, 8)
        print(f'{nseg} segment(s) per trial, {T // 32} chunks; per work item, microseconds (100 MHz wall clock):')
        for s_ in range(nseg):
            n = max(st[s_, 3], 1.0)
            print(f'  segment {s_}: items {int(st[s_, 3])}  entry -> state ready {st[s_, 0] / n / 100:7.1f}   steps {st[s_, 1] / n / 100:7.1f}   save + publish {st[s_, 2] / n / 100:6.1f}   (entry -> predecessor seen {st[s_, 4] / n / 100:5.1f}, state restored after another {st[s_, 5] / n / 100:5.1f})')
    sys.exit(0)
if '--fpi' in sys.argv:                                            # MCKF fixed-point branch by phase (-DUVS_FPI_STAMPS build), whole trials, alpha from argv
    alpha = float(sys.argv[sys.argv.index('--fpi') + 1]) if len(sys.argv) > sys.argv.index('--fpi') + 1 else 1.0
    T, K = 65536, 299
    cfg = bench.config2()
    cfg['noise']['noise_params']['alpha'] = alpha
    plan = uvs_amd.batch.plan_trials(cfg, cells=[alpha])
    fp = uvs_amd.engine.make_params(8, 6, 'MCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 2)
    fp.reserved = 1 << 8                                           # whole trials: one work item per chunk writes the stamps
    noise = uvs_amd.batch.device_noise(cfg, plan, 0, T, K, 'cuda', share=False)
    q0 = torch.as_tensor(plan.q_start, device='cuda')
    plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    for _ in range(2):
        out = uvs_amd.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
    torch.cuda.synchronize()
    stats = out['stats'].cpu().numpy().ravel()
    waves = T // 32
    st = np.stack([stats[96 * w: 96 * w + 8] for w in range(waves)])
    names = ['slot assignment + pull-in', 'park owners blocks + undo + factor', 'park predicted block + passes', 'commit + write-back + X stream', 'pull-back', '(unused)']
    fir = st[:, 6]
    print(f'alpha {alpha}: wavefronts {waves}, firings per wavefront mean {fir.mean():.1f} (max {fir.max():.0f}) of {K} steps; step loop {st[:, 7].mean():.0f} cycles per wavefront')
    tot = st[:, :5].sum(axis=1)
    print(f'  branch total {tot.sum() / fir.sum():.0f} cycles per firing = {100 * tot.mean() / st[:, 7].mean():.1f} % of the step loop')
    for i in range(5):
        print(f'  {names[i]:40s} {st[:, i].sum() / fir.sum():8.0f} cycles per firing')
    sys.exit(0)
alpha = float(sys.argv[sys.argv.index('--alpha') + 1]) if '--alpha' in sys.argv else None      # --alpha A: the sweep's alpha-stable noise instead of N(0, 1)
pos = [a for a in sys.argv[1:] if not a.startswith('--') and (sys.argv[sys.argv.index(a) - 1] != '--alpha')]
lanes = int(pos[0]) if pos else 2
T, K = 65536, 299
cfg = bench.config2()
plan = uvs_amd.batch.plan_trials(cfg, cells=[alpha or 1.5])
fp = uvs_amd.engine.make_params(8, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, lanes)
if alpha is None:
    noise = torch.empty((K, 8, T), dtype=torch.float64, device='cuda').normal_()
else:
    cfg['noise']['noise_params']['alpha'] = alpha
    noise = uvs_amd.batch.device_noise(cfg, plan, 0, T, K, 'cuda', share=False)
q0 = torch.as_tensor(plan.q_start, device='cuda')
plant = uvs_amd.SyntheticPlant.ur10().to_struct()
for _ in range(2):
    out = uvs_amd.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
torch.cuda.synchronize()
stats = out['stats'].cpu().numpy().ravel()
tpw = 64 // lanes
waves = T // tpw
st = np.stack([stats[3 * w * tpw: 3 * w * tpw + 8] for w in range(waves)])
names = ['noise-issue + plant', 'row updates', 'control law (QR)', 'logs + stats', '(100 MHz wall ticks)', 'loop edge', 'vmcnt(0) wait before the rows', '(unused)']
rt = st[:, 4].copy()
st[:, 4] = 0
tot = st.sum(axis=1)
print(f'waves {waves}, cycles per wave: mean {tot.mean():.0f}  min {tot.min():.0f}  max {tot.max():.0f};  per step {tot.mean() / K:.0f}')
print(f'shader clock while the kernel runs: {tot.mean() / (rt.mean() / 100e6) / 1e9:.3f} GHz  (loop wall time {rt.mean() / 100e6 * 1e3:.3f} ms per wavefront)')
for i, n in enumerate(names):
    if i == 4:
        continue
    print(f'  {n:24s} {st[:, i].mean() / K:9.0f} cycles/step  {100 * st[:, i].mean() / tot.mean():5.1f}%')

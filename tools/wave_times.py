#!/usr/bin/env python3
"""Where the launch time of a closed-loop kernel goes, wavefront by wavefront (diagnostic build -DUVS_WAVE_TIMES):
start / end wall time (100 MHz), XCD, CU and SIMD of every wavefront of one BASELINE config-2 launch per estimator.
usage (GPU box): make -C uncalibrated-visual-servoing_amd/csrc quick QDEF=-DUVS_WAVE_TIMES QOUT=../../tools/diag/libuvs_wt.so
                 UVS_LIB_PATH=tools/diag/libuvs_wt.so python tools/wave_times.py [--methods ...] [--alpha A]"""
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
ap.add_argument('--alpha', type=float, default=1.5)
ap.add_argument('--lanes', type=int, default=0)
ap.add_argument('--map', type=int, default=0, help='workgroup -> trial chunk mapping of the diagnostic build (fp.reserved)')
ap.add_argument('--want', default='x,err,q')
ap.add_argument('--segments', type=int, default=0, help='MCKF: work items per trial chunk (bits 8-15 of fp.reserved); every item is stamped')
args = ap.parse_args()
T, dev = args.trials, torch.device('cuda')
cfg = bench.config2()
cfg['experiments']['epoch'] = T
cfg['noise']['noise_params']['alpha'] = args.alpha
K = len(engine.loop_clock(0.05, 15))
plan = batch.plan_trials(cfg, cells=[args.alpha])
noise = batch.device_noise(cfg, plan, 0, T, K, dev, share=False)
q0 = torch.as_tensor(plan.q_start.copy(), device=dev)
plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
tpw = 64 // (args.lanes or 2)
for meth in args.methods.split(','):
    fp = engine.make_params(8, 6, meth, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, args.lanes)
    fp.reserved = (args.map << 16) | (args.segments << 8)
    import ctypes as C
    nseg = int(uvs_amd.lib().uvs_rmckf_closed_loop_segments(C.byref(fp), C.byref(plant), T))
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = engine.closed_loop(fp, plant, q0, noise, want=tuple(w for w in args.want.split(',') if w))
        e1.record(); torch.cuda.synchronize()
    st = out['stats'].cpu().numpy().ravel()
    nw = T // tpw
    w = np.stack([st[3 * i * tpw + 4 * sg: 3 * i * tpw + 4 * sg + 4] for sg in range(nseg) for i in range(nw)])
    nw *= nseg
    if nseg > 1:                                                   # per segment, and the chains of the chunks that end last
        nchunk = nw // nseg
        W = w.reshape(nseg, nchunk, 4)
        base = W[:, :, 0].min()
        S0, S1 = (W[:, :, 0] - base) / 100.0, (W[:, :, 1] - base) / 100.0
        for sg in range(nseg):
            d = S1[sg] - S0[sg]
            print(f'   segment {sg}: starts p50 {np.median(S0[sg]):.0f} us, durations p10 {np.quantile(d, .1):.0f} p50 {np.median(d):.0f} p90 {np.quantile(d, .9):.0f} p99 {np.quantile(d, .99):.0f} max {d.max():.0f} us')
        for c in np.argsort(S1[-1])[-4:]:
            print(f'   chunk {c}: ' + '  '.join(f'[{S0[sg, c]:.0f} .. {S1[sg, c]:.0f}]' for sg in range(nseg)) + f'  busy {np.sum(S1[:, c] - S0[:, c]):.0f} us')
    w = w[np.argsort(w[:, 0])]
    t0, t1 = (w[:, 0] - w[:, 0].min()) / 100.0, (w[:, 1] - w[:, 0].min()) / 100.0          # microseconds
    hw, xcc = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    dur = t1 - t0
    span = t1.max()
    print(f'== {meth}: event time {e0.elapsed_time(e1):.3f} ms; wavefronts {nw}; first start .. last end {span / 1e3:.3f} ms; residency (sum of durations / slots x span) '
          f'{dur.sum() / (span * min(nw, 1024 * (2 if meth in ("KF", "IMCCKF") else 1))):.3f}; work items per chunk {nseg}')
    print(f'   start times us: p0 {t0.min():.0f} p50 {np.median(t0):.0f} p90 {np.quantile(t0, .9):.0f} max {t0.max():.0f};  later-round starts (> 100 us): {(t0 > 100).sum()}')
    print(f'   durations  us: min {dur.min():.0f} p10 {np.quantile(dur, .1):.0f} p50 {np.median(dur):.0f} p90 {np.quantile(dur, .9):.0f} max {dur.max():.0f}')
    print(f'   end times  us: p10 {np.quantile(t1, .1):.0f} p50 {np.median(t1):.0f} p90 {np.quantile(t1, .9):.0f} p99 {np.quantile(t1, .99):.0f} max {t1.max():.0f}')
    for x in range(8):
        m = xcc == x
        if m.any():
            first = m & (t0 < 100)
            print(f'   XCD {x}: {m.sum():5d} wavefronts, mean duration {dur[m].mean():7.0f} us (first round {dur[first].mean():7.0f}), last end {t1[m].max():7.0f} us, CUs seen {len(set(zip(se[m], sh[m], cu[m])))}')
    key = xcc * 4096 + se * 256 + sh * 16 + cu
    per_cu = np.bincount(key)
    per_cu = per_cu[per_cu > 0]
    print(f'   wavefronts per CU: min {per_cu.min()} max {per_cu.max()} over {len(per_cu)} CUs')

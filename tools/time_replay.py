#!/usr/bin/env python3
"""Replay-mode kernel microbenchmark: T trials x K steps of synthetic consistent streams through
uvs_rmckf_replay_f64; reports updates/s and algorithmic GB/s for the requested outputs.  Run on the GPU box."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import uvs_amd
from uvs_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument('--trials', type=int, default=65536)
ap.add_argument('--steps', type=int, default=299)
ap.add_argument('--lanes', type=int, default=0)
ap.add_argument('--method', default='GMCKF')
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--layout', default='kct', choices=['kct', 'ktc'], help='physical layout of the output streams')
ap.add_argument('--in-layout', default='kct', choices=['kct', 'ktc'], help='physical layout of the f / dq streams')
ap.add_argument('--inplace', action='store_true', help='diagnostic: every step overwrites the rows of step 0 (output step stride 0): stores without HBM write traffic')
args = ap.parse_args()
T, K, m, n = args.trials, args.steps, 8, 6
dev = 'cuda'
g = torch.Generator(device=dev); g.manual_seed(1)
# consistent streams: f_{k+1} = f_k + J dq_k dt + noise, J ~ interaction-matrix scale; layout [step][component][trial]
J = torch.randn((m, n, T), device=dev, dtype=torch.float64, generator=g) * 50
dq = torch.randn((K, n, T), device=dev, dtype=torch.float64, generator=g) * 0.2
f = torch.empty((K + 1, m, T), device=dev, dtype=torch.float64)
f[0] = 128 + 20 * torch.randn((m, T), device=dev, dtype=torch.float64, generator=g)
for k in range(K):
    f[k + 1] = f[k] + torch.einsum('mnt,nt->mt', J, dq[k]) * 0.05 + torch.randn((m, T), device=dev, dtype=torch.float64, generator=g)
x0 = (J + 5 * torch.randn(J.shape, device=dev, dtype=torch.float64, generator=g)).permute(2, 0, 1).reshape(T, m * n).contiguous()
fp = engine.make_params(m, n, args.method, 10.0, False, 0.05, 15, 0.2, [128.0] * m, False, args.lanes, K)
import ctypes as C
NV = uvs_amd._lib.NULL_VIEW
status = torch.zeros(T, dtype=torch.int32, device=dev)
k_done = torch.zeros(T, dtype=torch.int32, device=dev)
bufs = {'x': engine.alloc_stream(T, K, m * n, args.layout, dev), 'err': engine.alloc_stream(T, K, m, args.layout, dev), 'dqcmd': engine.alloc_stream(T, K, n, args.layout, dev)}
if args.in_layout == 'ktc':
    f, dq = f.permute(0, 2, 1).contiguous(), dq.permute(0, 2, 1).contiguous()
flat = lambda t: uvs_amd._lib.View(t.data_ptr(), t.stride(0), 0, t.stride(1))
for want, nbytes in ((('x', 'err', 'dqcmd'), 8 * (m + n + m * n + m + n)), (('x', 'err'), 8 * (m + n + m * n + m)), ((), 8 * (m + n))):
    v = {k: (engine.stream_view(bufs[k], args.layout) if k in want else NV) for k in bufs}
    if args.inplace:
        v = {k: (uvs_amd._lib.View(x.base, x.trial_stride, 0, x.comp_stride) if x.base else x) for k, x in v.items()}
    times = []
    for _ in range(args.reps + 1):                                 # buffers preallocated: the kernel alone, as bench.py times it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = uvs_amd.lib().uvs_rmckf_replay_f64(C.byref(fp), T, engine.stream_view(f, args.in_layout), engine.stream_view(dq, args.in_layout), flat(x0), v['x'], v['err'], NV,
                                                v['dqcmd'], status.data_ptr(), k_done.data_ptr(), NV, NV, engine._stream())
        uvs_amd._lib.check(rc)
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    best = min(times[1:])
    print(f'{args.method} lanes={args.lanes} out={args.layout} in={args.in_layout} outputs={want}: {best:.3f} ms  {T * K / best / 1e6:.2f} G updates/s  {T * K * nbytes / best / 1e9:.2f} TB/s algorithmic ({nbytes} B/update)  failed {int((status != 0).sum())}')

#!/usr/bin/env python3
"""Where does on-device noise generation spend its time?  (host seeding, upload, kernel) -- run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import uvs_amd
from uvs_amd import noise_device as nd, pcg, engine

T, K, m = 65536, 299, 8
cases = [('ALPHA_STABLE', dict(alpha=1.5, beta=0, gamma=1, delta=0)), ('ALPHA_STABLE', dict(alpha=1.0, beta=0, gamma=1, delta=0)),
         ('ALPHA_STABLE', dict(alpha=2.0, beta=0, gamma=1, delta=0)), ('GAUSSIAN_MIXTURE', dict(std=1.0, mean=50.0, rho=0.1)),
         ('WHITE_NOISE', dict(std=1.0))]
seeds = 123456 + np.arange(T)
for name, p in cases:
    nt = uvs_amd.NoiseType[name]
    t0 = time.perf_counter()
    gs = nd.generator_seeds(nt, seeds, m)
    states = pcg.pcg64_states(gs)
    t1 = time.perf_counter()
    st_dev = torch.as_tensor(states.view(np.int64), device='cuda'); torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = engine.alloc_stream(T, K, m, 'kct', 'cuda')
    for rep in range(3):
        torch.cuda.synchronize(); t3 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import ctypes as C
        q = nd.make_noise_params(nt, p, m, K)
        e0.record()
        rc = uvs_amd._lib.lib().uvs_noise_generate_f64(C.byref(q), T, st_dev.data_ptr(), nd._zig('cuda').data_ptr(), engine.stream_view(out, 'kct'), engine._stream())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    nd.generate(nt, p, seeds, m, K, out=out); torch.cuda.synchronize()
    whole = time.perf_counter() - t4
    print(f'{name:18s} {p}: generate() end to end {1e3 * whole:6.1f} ms | host seeding (pcg.py, no longer on the path) {1e3*(t1-t0):7.1f} ms  upload {1e3*(t2-t1):6.1f} ms  kernel {ms:7.2f} ms  ({T*K*m/ms/1e6:.1f} G samples/s, {T*K*m*8/ms/1e6:.0f} GB/s)')

#!/usr/bin/env python3
"""Where the 63 us of a drop-in step go (engine.FilterBank.step_host): host time of the call without the synchronisation, kernel duration by
HIP events, the synchronisation, and the numpy copies around it; T = 1 and T = 64.  usage (GPU box): python tools/time_step_route.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from uvs_amd import engine  # noqa: E402
import bench  # noqa: E402

des = bench.config2()['experiments']['desired_f']
rng = np.random.default_rng(7)
for T in (1, 64):
    fp = engine.make_params(8, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, des, True, 0, 0)
    bank = engine.FilterBank(fp, T, rng.normal(size=(T, 48)) * 50, 'cuda')
    f = np.asarray(des)[None] + rng.normal(size=(T, 8))
    dq = np.zeros((T, 6))
    whole, kern = [], []
    for k in range(450):
        f_old, f = f, f + 0.1 * rng.normal(size=(T, 8))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if k % 2:                                                     # every other call with events around the launch (they cost a few us themselves)
            e0.record()
        dq_h, err_h, _, st = bank.step_host(f, f_old, k % 299, dq)
        if k % 2:
            e1.record()
            e1.synchronize()
            kern.append(e0.elapsed_time(e1) * 1e3)
        else:
            whole.append((time.perf_counter() - t0) * 1e6)
        dq = dq_h.copy()
    # the same call with the synchronisation left out: launch cost on the host
    h = bank._host
    stream = torch.cuda.current_stream()
    import ctypes as C
    t0 = time.perf_counter()
    for k in range(200):
        i = h['calls'] & 1
        h['call'][i](0, k % 299, h['ptr']['dq'][1 - i], C.c_void_p(stream.cuda_stream))
        h['calls'] += 1
    t1 = time.perf_counter()
    stream.synchronize()
    t2 = time.perf_counter()
    back_to_back = (t2 - t0) / 200 * 1e6
    print(f'T = {T:2d}: step_host median {np.median(whole[50:]):.1f} us;  stream-ordered duration of one step (events around the launch, includes the launch gap) '
          f'median {np.median(kern[25:]):.1f} us;  host side of the launch alone {(t1 - t0) / 200 * 1e6:.1f} us per call;  200 steps back to back without '
          f'synchronising in between {back_to_back:.1f} us per step (kernel-bound)')

#!/usr/bin/env python3
"""Do the closed-loop kernel (one 294-register wavefront per SIMD) and the noise generator (84 registers, no LDS) share the SIMDs when
they run on two streams?  Wall time of R closed-loop launches + R noise launches, back to back on one stream against on two streams.
usage (GPU box): python tools/overlap_probe.py [--reps R]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import uvs_amd
from uvs_amd import engine, batch, noise_device as nd
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=6)
ap.add_argument('--alpha', type=float, default=1.5)
args = ap.parse_args()
T, dev = 65536, torch.device('cuda')
cfg = bench.config2()
cfg['experiments']['epoch'] = T
K = len(engine.loop_clock(0.05, 15))
plan = batch.plan_trials(cfg, cells=[args.alpha])
noise = batch.device_noise(cfg, plan, 0, T, K, dev)
q0 = torch.as_tensor(plan.q_start.copy(), device=dev)
plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
fp = engine.make_params(8, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0)
nt = uvs_amd.NoiseType.ALPHA_STABLE
states = nd.device_generator_states(nt, plan.seed[:T], 8, dev)
q = nd.make_noise_params(nt, dict(alpha=args.alpha, beta=0, gamma=1, delta=0), 8, K)
out2 = engine.alloc_stream(T, K, 8, 'kct', dev)
zig = nd._zig(dev)
keep = {}


def loop():
    keep['o'] = engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))


def gen():
    s = torch.cuda.current_stream().cuda_stream
    uvs_amd._lib.check(uvs_amd.lib().uvs_noise_generate_f64(C.byref(q), T, states.data_ptr(), zig.data_ptr(), engine.stream_view(out2, 'kct'), C.c_void_p(s)))


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


R = args.reps
side = torch.cuda.Stream(device=dev)
hi = torch.cuda.Stream(device=dev, priority=-1)


def only_loop():
    for _ in range(R): loop()


def only_gen():
    for _ in range(R): gen()


def serial():
    for _ in range(R): gen(); loop()


def two_streams():
    for _ in range(R):
        with torch.cuda.stream(side): gen()
        loop()
    torch.cuda.current_stream().wait_stream(side)


def two_streams_loop_first():
    for _ in range(R):
        loop()
        with torch.cuda.stream(side): gen()
    torch.cuda.current_stream().wait_stream(side)


def two_streams_gen_high_priority():
    for _ in range(R):
        with torch.cuda.stream(hi): gen()
        loop()
    torch.cuda.current_stream().wait_stream(hi)


for name, fn in (('closed loop alone', only_loop), ('noise alone', only_gen), ('one stream: noise, closed loop, ...', serial), ('two streams', two_streams),
                 ('two streams, closed loop enqueued first', two_streams_loop_first), ('two streams, noise on a high-priority stream', two_streams_gen_high_priority)):
    ms = min(timed(fn) for _ in range(3))
    print(f'{name:50s} {ms / R:7.3f} ms per (cell)', flush=True)

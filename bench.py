#!/usr/bin/env python3
"""Headline benchmark: RMCKF updates/s over a Monte-Carlo batch (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: the closed-loop kernel advancing every trial of BASELINE
config 2 (4 features / 6 DoF, GMCKF = RMCKF, sigma 10, alpha-stable noise alpha = 1.5, 65 536 trials, 299 filter
updates per trial) from q_start to the end of the trial, with the noise streams already resident in HBM and the
per-step X / error / joint logs written to HBM.  One *update* = predict + correntropy-weighted correct + control law of
one filter for one time step.

Multi-GPU (one process per GPU under torch.distributed.run): trials are enumerated globally (main.py:121-139: seed0 + t, t-th jitter
draw) and rank r owns the contiguous shard dist.shard_range(total, r, N) -- no data-path collective -- followed by one RCCL all-gather
of the per-trial [ISE, IAE, ITAE, status] rows, which is inside the timed region.  --scaling weak (default: per-GPU work fixed, the
driver's scaling series) gives every GPU the config's size; --scaling strong keeps the TOTAL at the config's size (north_star's series, "a
65 536-trial batch at 1, 2, 4 and 8 MI355X") -- a weak config-2 run on N > 1 ranks times that series too, right after the timed region, and
reports it as the side object multi_gpu.strong_series; --config 4 is BASELINE config 4: 1 048 576 trials in total over however many ranks
there are (always strong).

Plain `python bench.py --gpus N` with N > 1 starts its own N ranks (fresh children under torch.distributed.run, before this process
touches the GPU) and relays rank 0's line; under torch.distributed.run it is one of the ranks.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields, incl. `roofline` and `cpu_baseline`).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')      # before numpy loads OpenBLAS: one BLAS thread per process
os.environ.setdefault('OMP_NUM_THREADS', '1')

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRIALS_PER_GPU = 65536
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
ALPHA = 1.5


def config2():
    """BASELINE config 2 in the reference's config.json schema."""
    return {
        'log_level': 'INFO',
        'experiments': {'dt': 0.05, 't_max': 15, 'epoch': TRIALS_PER_GPU, 'ibvs_gain': 0.2,
                        'q_start': [0.0, 0.0, 1.96349541, 0.0, -1.57079633, 0],
                        'desired_f': [149.0, 145.0, 125.0, 121.0, 101.0, 145.0, 125.0, 169.0],
                        'visualization': False, 'change_q_start': True, 'seed': 12345},
        'estimator': {'method': 'GMCKF', 'estimator_params': {'initial_guess': True, 'kernel_bw': 10, 'fpi_threshold': 0.1,
                                                              'fpi_epoch_max': 1000, 'annealing': False}},
        'noise': {'type': 'ALPHA_STABLE', 'noise_params': {'alpha': ALPHA, 'beta': 0, 'gamma': 1, 'delta': 0},
                  'hold': False, 'hold_time': 0.5, 'seed': 123456},
    }


# ---------------------------------------------------------------------------------------------- host-side input generation
def _noise_chunk(args):
    import uvs_amd
    seeds, steps = args
    return uvs_amd.noise_batch(uvs_amd.NoiseType.ALPHA_STABLE, dict(alpha=ALPHA, beta=0, gamma=1, delta=0), seeds, 8, steps)


def host_noise(seeds, steps, workers):
    """[steps][8][T] trial-fastest noise buffer, identical to NoiseProfiler streams (numpy PCG64), built on `workers` cores."""
    chunks = np.array_split(np.asarray(seeds), max(1, len(seeds) // 512))
    out = np.empty((steps, 8, len(seeds)))
    pos = 0
    if workers > 1:
        with mp.get_context('fork').Pool(workers) as pool:
            for block in pool.imap(_noise_chunk, [(c, steps) for c in chunks]):
                out[:, :, pos:pos + len(block)] = block.transpose(1, 2, 0)
                pos += len(block)
    else:
        for c in chunks:
            block = _noise_chunk((c, steps))
            out[:, :, pos:pos + len(block)] = block.transpose(1, 2, 0)
            pos += len(block)
    return out


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle, "port")
def _cpu_trial(args):
    os.environ['OPENBLAS_NUM_THREADS'] = '1'
    from oracle import noise_ref, plant_ref, rmckf_dense
    seed, q_start = args
    cfg = config2()
    ex = cfg['experiments']
    stream = noise_ref.NoiseStreamRef(8, noise_ref.ALPHA_STABLE, seed, alpha=ALPHA, beta=0, gamma=1, delta=0)
    noise = stream.take(299)                                      # noise generation is not part of an "update": pre-draw it
    it = iter(noise)
    t0 = time.perf_counter()
    out = rmckf_dense.run_closed_loop(plant_ref.PinholeUR10(ex['dt']), q_start, ex['desired_f'], lambda: next(it), ex['dt'],
                                      ex['t_max'], ex['ibvs_gain'], method='GMCKF', kernel_bw=10, annealing=False)
    return out['k_done'], time.perf_counter() - t0


def _cpu_trial_block(args):
    """Same trial through the per-row (block) numpy restatement, oracle/rmckf_block.py (SURVEY 8d asks for it beside the dense port)."""
    os.environ['OPENBLAS_NUM_THREADS'] = '1'
    from oracle import noise_ref, plant_ref, rmckf_block, rmckf_dense
    seed, q_start = args
    cfg = config2()
    ex = cfg['experiments']
    noise = noise_ref.NoiseStreamRef(8, noise_ref.ALPHA_STABLE, seed, alpha=ALPHA, beta=0, gamma=1, delta=0).take(299)
    discs = plant_ref.place_discs()
    robot = plant_ref.PinholeUR10(ex['dt'])
    robot.start(q_start)
    x0 = rmckf_dense.analytic_initial_guess(robot, robot.features(), 8, 6)
    t0 = time.perf_counter()
    out = rmckf_block.run_closed_loop(lambda q: plant_ref.project(plant_ref.fkine_all(q)[5], discs), q_start, ex['desired_f'], noise, ex['dt'],
                                      ex['t_max'], ex['ibvs_gain'], x0, method='GMCKF', kernel_bw=10.0, annealing=False)
    return out['k_done'], time.perf_counter() - t0


def available_cores():
    """Cores this process may really use: the scheduler affinity, cut down to the cgroup CPU quota when the container has one
    (a 256-way affinity mask over a 16-core quota would otherwise report 256 'cores' running at a sixteenth of their speed)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]            # cgroup v2
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                                                          # cgroup v1
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                cores = max(1, min(cores, quota // period))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(q_starts, budget_trials_per_core=256):
    """Dense numpy restatement of the reference loop (oracle/rmckf_dense.py, op-for-op experiment.py:125-343) on all host cores."""
    os.environ['OPENBLAS_NUM_THREADS'] = '1'
    cores = available_cores()
    n_trials = min(len(q_starts), budget_trials_per_core * cores)
    jobs = [(123456 + i, q_starts[i]) for i in range(n_trials)]
    t0 = time.perf_counter()
    with mp.get_context('fork').Pool(cores) as pool:
        res = pool.map(_cpu_trial, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    updates = sum(r[0] for r in res)
    single = updates / sum(r[1] for r in res)                     # per-process rate (what one reference process achieves)
    nb = min(len(jobs), max(cores, n_trials // 4))                # the block form is ~3x slower per trial in numpy (more, smaller calls)
    t0 = time.perf_counter()
    with mp.get_context('fork').Pool(cores) as pool:
        res_b = pool.map(_cpu_trial_block, jobs[:nb], chunksize=1)
    wall_b = time.perf_counter() - t0
    block = {'value': sum(r[0] for r in res_b) / wall_b, 'unit': 'updates/s', 'cores': cores,
             'sample': f'{nb} trials x 299 steps of config 2 through oracle/rmckf_block.py (per-row numpy form, same cores)',
             'per_process_updates_per_s': sum(r[0] for r in res_b) / sum(r[1] for r in res_b)}
    return {'value': updates / wall, 'unit': 'updates/s', 'cores': cores, 'kind': 'port', 'block_form': block,
            'sample': f'{n_trials} trials x 299 steps of config 2 through oracle/rmckf_dense.py (dense numpy, OPENBLAS_NUM_THREADS=1, '
                      f'one process per core, noise pre-drawn); os.cpu_count()={os.cpu_count()}, affinity={len(os.sched_getaffinity(0))}, usable (cgroup quota)={cores}',
            'per_process_updates_per_s': single}


# ---------------------------------------------------------------------------------------------- main
def replay_side_measurement(torch, engine, uvs_amd, fp_closed, plant, q_start, x_buf, err_buf, T, K, M, N, reps=20):
    """Replay mode (SURVEY 8d kernel microbenchmark): the estimator (+ control law) over recorded feature / joint-delta streams,
    the I/O north_star prices: read f and dq, write X and err.  Inputs per SURVEY 8d "replay-mode synthetic inputs", built on the device:
    per trial t, J_true = the analytic initial-guess matrix at the trial's jittered q_start (one 1-step launch of the closed loop returns it
    together with the noise-free features f_0); dq_k = dq_0 exp(-lambda k dt) + N(0, 1e-3^2) with dq_0 ~ N(0, 0.2^2); f_k = f_{k-1} +
    J_true dq_k dt, observed as f_k + noise_k.  Random streams: the library's NoiseProfiler-compatible generator seeded 987654 + t
    (white noise for dq_0 and the dq jitter, alpha-stable alpha = 1.5 for the observation noise) -- numpy PCG64 streams of seed
    987654 + t + 10 i (noise.py:66-70), not one PCG64(987654 + t) per trial: 65 536 host generators would cost the bench minutes.
    Reuses the closed-loop run's output buffers.  Reported next to, never as, the closed-loop `value`."""
    import ctypes as C
    dev = x_buf.device
    NT = uvs_amd.NoiseType
    seeds = np.arange(T, dtype=np.uint64) + np.uint64(987654)
    lam, dt = fp_closed.gain, fp_closed.dt
    fp1 = engine.make_params(M, N, 'GMCKF', fp_closed.kernel_bw, False, dt, dt * fp_closed.k_max, lam, list(fp_closed.desired)[:M], True, 0, 1)
    first = engine.closed_loop(fp1, plant, q_start, None, want=('f',), final_state=True)
    J = first['x_final'].view(T, M, N).permute(1, 2, 0).contiguous()                 # [m][n][trial]
    f0 = first['f'][0].clone()                                                       # [m][trial]
    del first
    white = uvs_amd.noise_device.generate(NT.WHITE_NOISE, dict(std=1.0), seeds, N, K + 1, layout='kct', device=dev)      # [K + 1][n][trial]
    decay = torch.exp(-lam * dt * torch.arange(K, device=dev, dtype=torch.float64))
    dq = 0.2 * white[0][None] * decay[:, None, None] + 1e-3 * white[1:]
    del white
    obs = uvs_amd.noise_device.generate(NT.ALPHA_STABLE, dict(alpha=ALPHA, beta=0, gamma=1, delta=0), seeds + np.uint64(1 << 20), M, K + 1, layout='kct', device=dev)
    f = torch.empty((K + 1, M, T), device=dev, dtype=torch.float64)
    clean = f0
    f[0] = clean + obs[0]
    for k in range(K):
        clean = clean + torch.einsum('mnt,nt->mt', J, dq[k]) * dt
        f[k + 1] = clean + obs[k + 1]
    del obs, clean
    g = torch.Generator(device=dev)
    g.manual_seed(987654)
    x0 = (J * (1 + 0.05 * torch.randn(J.shape, device=dev, dtype=torch.float64, generator=g))).permute(2, 0, 1).reshape(T, M * N).contiguous()
    del J
    fp = engine.make_params(M, N, 'GMCKF', fp_closed.kernel_bw, bool(fp_closed.annealing), fp_closed.dt, fp_closed.dt * fp_closed.k_max, fp_closed.gain,
                            list(fp_closed.desired)[:M], False, 0, K)
    cmd_buf = engine.alloc_stream(T, K, N, 'kct', dev)
    status = torch.zeros(T, dtype=torch.int32, device=dev)
    k_done = torch.zeros(T, dtype=torch.int32, device=dev)
    NV = uvs_amd._lib.NULL_VIEW
    flat = uvs_amd._lib.View(x0.data_ptr(), x0.stride(0), 0, x0.stride(1))
    out = {}
    for name, cmd, b_alg, lay in (('estimator_and_control_law', True, 8 * (2 * M + 2 * N + M * N), 'kct'), ('estimator_only', False, 8 * (2 * M + N + M * N), 'kct'),
                                  ('estimator_only_records', False, 8 * (2 * M + N + M * N), 'ktc')):
        if lay == 'ktc':                                          # the same streams as per-trial records ([step][trial][component]); same buffers
            f, dq = f.permute(0, 2, 1).contiguous(), dq.permute(0, 2, 1).contiguous()
            dense = lambda t, c: (t._base if t._base is not None else t).reshape(-1)[:K * T * c].view(K, T, c)   # noqa: E731 -- (pitched rows: the allocation behind the view)
            x_buf, err_buf = dense(x_buf, M * N), dense(err_buf, M)
        ms = []
        for _ in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = uvs_amd.lib().uvs_rmckf_replay_f64(C.byref(fp), T, engine.stream_view(f, lay), engine.stream_view(dq, lay), flat,
                                                    engine.stream_view(x_buf, lay), engine.stream_view(err_buf, lay), NV,
                                                    engine.stream_view(cmd_buf, 'kct') if cmd else NV, status.data_ptr(), k_done.data_ptr(), NV, NV,
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
            uvs_amd._lib.check(rc)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        avg = float(np.mean(ms[1:]))
        updates = int(k_done.sum().item())
        gbs = updates * b_alg / (avg * 1e-3) / 1e9
        out[name] = {'updates_per_s': updates / (avg * 1e-3), 'avg_kernel_ms': avg, 'algorithmic_bytes_per_update': b_alg, 'achieved': gbs, 'unit': 'GB/s',
                     'frac': gbs / HBM_PEAK_GBS, 'failed_trials': int((status != 0).sum().item())}
    # ---- the single-precision measured variant: the same streams rounded to fp32, fp32 state; lower precision, never `value`
    if lay == 'ktc':
        f, dq = f.permute(0, 2, 1), dq.permute(0, 2, 1)                                  # back to [step][component][trial]
    f, dq = f.contiguous(), dq.contiguous()
    f32, dq32, x032 = f.to(torch.float32), dq.to(torch.float32), x0.to(torch.float32).contiguous()
    x64 = engine.alloc_stream(T, 8, M * N, 'kct', dev)                                # fp64 X of the first 8 steps for the deviation figure
    fp8 = engine.make_params(M, N, 'GMCKF', fp_closed.kernel_bw, bool(fp_closed.annealing), fp_closed.dt, fp_closed.dt * fp_closed.k_max, fp_closed.gain,
                             list(fp_closed.desired)[:M], False, 0, 8)
    uvs_amd._lib.check(uvs_amd.lib().uvs_rmckf_replay_f64(C.byref(fp8), T, engine.stream_view(f, 'kct'), engine.stream_view(dq, 'kct'), flat,
                                                          engine.stream_view(x64, 'kct'), NV, NV, NV, status.data_ptr(), k_done.data_ptr(), NV, NV,
                                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    del f, dq
    x32 = torch.empty((K, M * N, T), dtype=torch.float32, device=dev)
    e32 = torch.empty((K, M, T), dtype=torch.float32, device=dev)
    flat32 = uvs_amd._lib.View(x032.data_ptr(), x032.stride(0), 0, x032.stride(1))
    ms = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = uvs_amd.lib().uvs_rmckf_replay_f32(C.byref(fp), T, engine.stream_view(f32, 'kct'), engine.stream_view(dq32, 'kct'), flat32, engine.stream_view(x32, 'kct'),
                                               engine.stream_view(e32, 'kct'), status.data_ptr(), k_done.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        uvs_amd._lib.check(rc)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    avg = float(np.mean(ms[1:]))
    updates = int(k_done.sum().item())
    b32 = 4 * (2 * M + N + M * N)
    gbs = updates * b32 / (avg * 1e-3) / 1e9
    dev8 = float(((x32[:8].double() - x64).abs().amax() / x64.abs().amax()).item())
    out['estimator_only_f32'] = {'updates_per_s': updates / (avg * 1e-3), 'avg_kernel_ms': avg, 'algorithmic_bytes_per_update': b32, 'achieved': gbs, 'unit': 'GB/s',
                                 'frac': gbs / HBM_PEAK_GBS, 'failed_trials': int((status != 0).sum().item()), 'dtype': 'f32',
                                 'x_rel_deviation_from_f64_first_8_steps': dev8,
                                 'note': 'LOWER PRECISION, measured variant only (SURVEY 8d): fp32 streams and fp32 state, v_pk_fma_f32; per-step X deviates ~1e-6 from the fp64 '
                                         'reference (tests/test_gpu_replay_f32.py bounds it at 1e-5, open loop); never `value`, no closed-loop path'}
    del x32, e32, f32, dq32, x64
    out['kernel'] = ('estimator_and_control_law: replay_rows_kernel<8,6,4,GMCKF,true,true,BYWAVE,CW=2> (4 estimator + 2 control-law wavefronts per 64 trials); estimator_only: replay_rows_kernel<8,6,4,GMCKF,true,true,BYWAVE> '
                     '(row groups of a filter in the 4 wavefronts of a 64-trial workgroup, 2 wavefronts/SIMD); estimator_only_records: replay_rows_kernel<...,REC> (4 lane groups, '
                     'LDS-transposed 1 KB stores)')
    out['streams'] = ('read f (m) + dq (n), write X (mn) + err (m) [+ commanded dq (n)] per update; [step][component][trial], estimator_only_records: per-trial records '
                      '[step][trial][component]')
    out['inputs'] = ('SURVEY 8d: J_true = analytic initial guess at the jittered q_start, dq_k = dq_0 exp(-lambda k dt) + N(0, 1e-3^2), f_k = f_{k-1} + J_true dq_k dt + '
                     'alpha-stable observation noise, X0 = J_true (1 + 5 % noise); NoiseProfiler-compatible device streams seeded 987654 + t')
    return out


def side_config(config, torch, uvs_amd, engine, batch, dev, trials=None, reps=5, warm=2, hold=False, lanes=0):
    """BASELINE configs 3 and 5 measured in the same run as side objects of the bench line (never as `value`): a few launches of the
    full-size workload, HIP events on the launch stream around every launch, inputs resident in HBM."""
    import ctypes as C
    cfg = config2()
    K = len(engine.loop_clock(0.05, 15))
    NV = uvs_amd._lib.NULL_VIEW
    flat = lambda t: uvs_amd._lib.View(t.data_ptr(), t.stride(0), 0, t.stride(1))     # noqa: E731
    if config == 3:
        T, M, N, layout = trials or 262144, 8, 6, 'kct'
        cfg['noise'].update(type='GAUSSIAN_MIXTURE', noise_params={'std': 1.0, 'mean': 50.0, 'rho': 0.1}, hold=bool(hold), hold_time=0.5)
        cfg['estimator']['estimator_params']['annealing'] = True
        cfg['experiments']['epoch'] = T
        plan = batch.plan_trials(cfg, cells=[0.1])
        noise = batch.device_noise(cfg, plan, 0, T, K, dev, share=False)
        q0 = torch.as_tensor(plan.q_start.copy(), device=dev)
        fp = engine.make_params(8, 6, 'GMCKF', 10, True, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, lanes)
        plant, x0 = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct(), None
        workload = f'BASELINE config 3: 4-feature UR10 closed loop, GMCKF(RMCKF) annealed sigma, Gaussian mixture rho=0.1 mean=50 hold={bool(hold)}, {T} trials x {K} updates, X+err+q logged'
        kernel = 'closed_loop_tuned_kernel<8,6,2,GMCKF,DH(axis-aligned UR10 table),2,true>'
    else:
        T, M, N, layout = trials or TRIALS_PER_GPU, 32, 7, 'ktc'
        cfg['experiments']['epoch'] = T
        plan = batch.plan_trials(cfg, cells=[ALPHA])
        lin = uvs_amd.LinearPlant.random(M, N, seed=2)
        rng = np.random.default_rng(5)
        q_goal = lin.q0 + rng.uniform(-0.3, 0.3, N)
        desired = lin.features(q_goal)
        q0 = torch.as_tensor(q_goal + np.random.default_rng(12345).uniform(-0.15, 0.15, (T, N)), device=dev)
        x0 = torch.as_tensor(np.tile((lin.J * (1 + 0.1 * rng.normal(size=lin.J.shape))).ravel(), (T, 1)), device=dev)
        noise = uvs_amd.noise_device.generate(uvs_amd.NoiseType.ALPHA_STABLE, dict(alpha=ALPHA, beta=0, gamma=1, delta=0), plan.seed, M, K, layout=layout, device=dev)
        fp = engine.make_params(M, N, 'GMCKF', 10, False, 0.05, 15, 0.2, desired, False, lanes)
        plant = lin.to_struct(dev)
        workload = f'BASELINE config 5: synthetic 16-feature / 7-DoF (m=32, n=7) linear plant, GMCKF(RMCKF) sigma=10, alpha-stable alpha=1.5, {T} trials x {K} updates, X+err+q logged, per-trial records'
        kernel = 'closed_loop_wide_kernel<32,7,8,GMCKF,true,true>'
    bufs = {k: engine.alloc_stream(T, K, c, layout, dev, zero=True) for k, c in (('x', M * N), ('err', M), ('q', N))}
    stats = torch.zeros((T, 3), dtype=torch.float64, device=dev)
    status = torch.zeros(T, dtype=torch.int32, device=dev)
    k_done = torch.zeros(T, dtype=torch.int32, device=dev)
    ms = []
    for i in range(warm + reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = engine.launch_closed_loop(fp, plant, T, flat(q0), engine.stream_view(noise, layout), NV if x0 is None else flat(x0),
            engine.stream_view(bufs['x'], layout), engine.stream_view(bufs['err'], layout), engine.stream_view(bufs['q'], layout), NV, NV,
            stats.data_ptr(), status.data_ptr(), k_done.data_ptr(), NV, NV, device=dev)
        uvs_amd._lib.check(rc)
        e1.record()
        torch.cuda.synchronize()
        if i >= warm:
            ms.append(e0.elapsed_time(e1))
    updates = int(k_done.sum().item())
    b_alg = 8 * (2 * M + N + M * N)
    avg = float(np.mean(ms))
    gbs = updates * b_alg / (avg * 1e-3) / 1e9
    out = {'workload': workload, 'trials': T, 'updates_per_launch': updates, 'launches_timed': reps, 'avg_kernel_ms': avg, 'min_kernel_ms': float(np.min(ms)),
           'updates_per_s': updates / (avg * 1e-3), 'algorithmic_bytes_per_update': b_alg, 'achieved': gbs, 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
           'frac': gbs / HBM_PEAK_GBS, 'kernel': kernel, 'layout': layout, 'failed_trials': int((status != 0).sum().item())}
    del bufs, noise
    torch.cuda.empty_cache()
    return out


CONFIG_TRIALS = {2: TRIALS_PER_GPU, 3: 262144, 4: 1048576, 5: TRIALS_PER_GPU}
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4        # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at the 2.4 GHz peak clock
FP64_VECTOR_PEAK_FLOPS = 78.6e12                     # MI355X_MICROARCH.md: fp64 vector peak


def drop_in_step_latency(torch, uvs_amd, engine, dev):  # noqa: C901
    """Latency of the drop-in STEP route (side object): what Experiment.run() pays per loop iteration for the estimator + control law when the
    robot is external (uncalibrated-visual-servoing_amd/experiment.py `_run_with_external_robot`, replacing experiment.py:166-312 of the
    reference, which measured 260-430 us per update on the build container's CPU).  A real servo trial of BASELINE config 2
    is driven from the host -- the package's host-side UR10 plant (plant.SyntheticRobot: kinematics and pinhole camera in numpy), the reference's
    noise stream, the analytic initial guess, the command of every step fed back to the joints -- and the host clock runs around
    FilterBank.step_host alone (numpy f / f_old in, numpy dq / err / status out, synchronised): T = 1 (the drop-in) and T = 64 copies of the trial,
    and the same trial through the round-5 route (three H2D tensors, launch, .item() + two .cpu() reads)."""
    cfg = config2()
    ex = cfg['experiments']
    des = np.asarray(ex['desired_f'], float)
    out = {}
    for T in (1, 64):
        for route in ('host_io', 'tensors'):
            robot = uvs_amd.SyntheticRobot(dt=ex['dt'])
            robot.start(ex['q_start'])
            prof = uvs_amd.NoiseProfiler(8, uvs_amd.NoiseType.ALPHA_STABLE, seed=cfg['noise']['seed'], noise_params=cfg['noise']['noise_params'])
            f = np.asarray(robot.features(), float)
            x0 = uvs_amd.Experiment._analytic_guess(robot, f, (256, 256), 8, 6)                # experiment.py:94-114
            fp = engine.make_params(8, 6, 'GMCKF', 10, False, ex['dt'], ex['t_max'], ex['ibvs_gain'], des, True, 0, 0)
            bank = engine.FilterBank(fp, T, np.tile(x0, (T, 1)), dev)
            dq = np.zeros((T, 6))
            lap = []
            for k in range(299):
                f_old = f
                f = np.asarray(robot.features(), float) + prof.getNoise()
                fT, foT = np.tile(f, (T, 1)), np.tile(f_old, (T, 1))
                t0 = time.perf_counter()
                if route == 'host_io':
                    dq_h, err_h, _, st = bank.step_host(fT, foT, k, dq)
                    bad = int(st[0])
                    dq = dq_h.copy()
                else:
                    to = lambda a: torch.as_tensor(a, device=dev)          # noqa: E731
                    dq_t, err_t, _, st = bank.step(to(fT), to(foT), to(dq), k)
                    bad = int(st[0].item())
                    dq = dq_t.cpu().numpy()
                    err_h = err_t.cpu().numpy()
                lap.append(time.perf_counter() - t0)
                assert bad == 0
                robot.setJointsPos(robot.getJointsPos() + dq[0] * ex['dt'])                  # experiment.py:320, 335
                robot.step()
            final_err = float(np.abs(np.asarray(robot.features()) - des).max())
            lap = np.array(lap[20:]) * 1e6
            out[f'T{T}_{route}'] = {'median_us': float(np.median(lap)), 'mean_us': float(lap.mean()), 'p95_us': float(np.quantile(lap, 0.95)),
                                     'final_feature_error_px': final_err}
    out = {'step_latency_us': {'T1': out['T1_host_io']['median_us'], 'T64': out['T64_host_io']['median_us']},
           'tensor_route_us': {'T1': out['T1_tensors']['median_us'], 'T64': out['T64_tensors']['median_us']}, 'detail': out}
    out['reference_us_per_update'] = [260, 430]
    out['note'] = ('host wall clock per Experiment-loop iteration of the estimator + control-law step on a real config-2 servo trial driven from the host (numpy plant, '
                   'reference noise stream), inputs and outputs as numpy arrays; host_io = FilterBank.step_host (pinned zero-copy records, one launch + one stream '
                   'synchronisation), tensors = the round-5 route; reference figure: BASELINE.md section 2 (numpy, build container)')
    return out


def parse_rocm_smi(text):
    """(package watts, shader clock in MHz, power cap in watts) from `rocm-smi --showpower --showclocks --showmaxpower` text; None where absent."""
    import re
    w = re.search(r'(?:Current Socket|Average) Graphics Package Power \(W\):\s*([\d.]+)', text)
    c = re.search(r'sclk clock level:\s*\S+\s*\((\d+)Mhz\)', text)
    m = re.search(r'Max Graphics Package Power \(W\):\s*([\d.]+)', text)
    return (float(w.group(1)) if w else None, int(c.group(1)) if c else None, float(m.group(1)) if m else None)


def power_under_load(torch, launch, device_index, max_seconds=10.0):
    """Package power and shader clock while the timed kernel runs back to back (side object, not part of the timed region): `rocm-smi`
    sampled up to three times from a thread, after 0.6 s of settling, while this thread keeps the launch queue full.  The closed-loop kernel
    is power-limited on MI355X: the clock it sustains, not the 2.4 GHz peak, is what its instruction rate sees.
    Returns None when rocm-smi is missing or prints nothing parseable."""
    import shutil
    import subprocess
    import threading
    exe = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    if not os.path.exists(exe):
        return None
    samples, cap = [], [None]

    def sampler():
        time.sleep(0.6)
        for _ in range(3):
            try:
                # rocm-smi is a `#!/usr/bin/env python3` script: run it with this interpreter (no env hop) and without any injected profiler library
                out = subprocess.run([sys.executable, os.path.realpath(exe), '-d', str(device_index), '--showpower', '--showclocks', '--showmaxpower'],
                                     capture_output=True, text=True, timeout=20, env=scrubbed_env()).stdout
            except Exception:                                      # noqa: BLE001 -- a missing / hanging tool must not cost the bench line
                return
            watts, mhz, cap_w = parse_rocm_smi(out)
            if cap_w is not None:
                cap[0] = cap_w
            if watts is not None and mhz is not None:
                samples.append((watts, mhz))

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0, launches = time.perf_counter(), 0
    while th.is_alive() and time.perf_counter() - t0 < max_seconds:
        for _ in range(10):
            launch()
        torch.cuda.synchronize()
        launches += 10
    th.join(timeout=30)
    if not samples:
        return None
    return {'package_watts': [w for w, _ in samples], 'sclk_mhz': [c for _, c in samples], 'cap_watts': cap[0], 'launches_while_sampling': launches,
            'note': 'rocm-smi --showpower --showclocks sampled while the timed kernel ran back to back after the timed region; not part of `value`'}


def e2e_sweep(torch, uvs_amd, engine, batch, dev, trials_per_cell=TRIALS_PER_GPU):
    """What main.py:104-196 does for one sweep, end to end on the GPU (side object, never `value`): the product's own sweep driver,
    batch.run_sweep -- for each of the reference's 12 cells alpha = linspace(1, 2, 12): device seeding + noise generation, closed loop
    with the per-step streams logged on the device, the per-trial [ISE, IAE, ITAE, status, k_done] rows copied to pinned host memory on
    a second stream -- at `trials_per_cell` trials per cell, timed by its host clock (synchronised on both sides)."""
    cfg = config2()
    cfg['experiments']['epoch'] = trials_per_cell
    variants = (('one_stream', dict(want=('x', 'err', 'q'))),                       # the headline's streams
                ('one_stream_csv_streams', dict(want=('err', 'q', 'f'))),           # what results.csv holds per step (main.py:152-194): 176 B per update
                ('stats_only', dict(want=())),                                      # per-trial rows only
                ('one_stream_every_stream_generated', dict(want=('x', 'err', 'q'), share_noise=False)))     # round 4's path: 8 T streams instead of T + 70
    out = {}
    plan = batch.plan_trials(cfg)                                  # once: planning 786 432 trials on the host takes 0.3 s, and a GPU left idle that long
    for name, kw in variants:                                      # starts the sweep at a low clock (profiles/r05/sweep_trace.txt)
        batch.run_sweep(cfg, device=dev, plan=plan, **kw)          # warm-up (allocator, tables)
        r = batch.run_sweep(cfg, device=dev, plan=plan, **kw)
        updates = int(r.k_done.sum())
        out[name] = {'wall_ms': r.seconds * 1e3, 'ms_per_cell': r.seconds * 1e3 / len(r.pieces), 'updates_per_s': updates / r.seconds,
                     'failed_trials': int((r.status != 0).sum())}
    # the pieces of one cell on their own (events on one stream), for the breakdown
    T, K, M = trials_per_cell, len(engine.loop_clock(0.05, 15)), 8
    fp = engine.make_params(M, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0)
    plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    q0 = torch.as_tensor(plan.q_start[5 * T:6 * T].copy(), device=dev)
    params = dict(alpha=float(plan.cells[5]), beta=0, gamma=1, delta=0)
    ev = lambda: torch.cuda.Event(enable_timing=True)             # noqa: E731
    res = None
    for _ in range(2):
        a, b, c_ = ev(), ev(), ev()
        a.record()
        view = uvs_amd.noise_device.generate_shared(uvs_amd.NoiseType.ALPHA_STABLE, params, int(plan.seed[5 * T]), T, M, K, device=dev)[1]
        b.record()
        res = engine.closed_loop(fp, plant, q0, view, want=('x', 'err', 'q'), reuse=res)
        c_.record()
        torch.cuda.synchronize()
    out['breakdown_one_cell_ms'] = {'seeding_and_noise_generation': a.elapsed_time(b), 'closed_loop_kernel': b.elapsed_time(c_)}
    out.update(workload=f'the reference sweep of main.py:104-148 through batch.run_sweep: 12 cells alpha = linspace(1, 2, 12) x {T} trials x {K} updates, GMCKF(RMCKF), '
                        'per-step streams logged on the device, per-trial [ISE, IAE, ITAE, status, k_done] rows copied to pinned host memory',
               cells=len(r.pieces), trials_per_cell=T, updates_total=updates,
               note='end to end, inputs NOT resident: includes device seeding and noise generation of every cell (the trial plan is made once, before: a sweep that '
                    'starts on an idle GPU spends its first cells at a lower clock, profiles/r05/sweep_trace.txt); never part of `value`.  '
                    'one_stream: X + err + q logged (the headline\'s streams); one_stream_csv_streams: err, q, f (what results.csv holds per step, '
                    '176 B per update); stats_only: no per-step stream; one_stream_every_stream_generated: without the stream aliasing (8 T streams)')
    del res, view
    torch.cuda.empty_cache()
    return out


def compact_line(line):
    """The one JSON line the driver stores has to stay small (round 4's 14 KB line was cut in the driver's record): prose and derivable fields
    leave it -- every `note` and the kernel / stream descriptions live in README.md "Reading the bench line" -- and nested numbers keep six
    significant digits.  Contract keys (top level, `config.workload`, `roofline`'s own fields, `cpu_baseline`) are untouched;
    UVS_BENCH_FULL_JSON=<path> writes the unabridged object."""
    prose = ('note', 'binds', 'kernel', 'streams', 'inputs', 'source', 'detail', 'launches_timed', 'noise_gen_workers', 'latency_option', 'h2d_inclusive_updates_per_s',
             'launches_while_sampling', 'trials_per_gpu')
    derivable = ('updates_per_s', 'updates_per_launch', 'algorithmic_bytes_per_update', 'unit', 'peak', 'wall_ms', 'updates_total', 'wave_instr_per_s',
                 'fp64_peak_flops', 'per_process_updates_per_s', 'min_kernel_ms', 'layout', 'cells', 'trials_per_cell')

    def walk(o, path):
        depth = len(path)
        if isinstance(o, dict):
            out = {}
            for k, v in o.items():
                top = path[0] if path else k
                if depth > 0 and top != 'cpu_baseline' and (k in prose or (k.endswith('_note') and k != 'backend_note')) and not (top == 'roofline' and depth == 1 and k == 'kernel'):
                    continue
                if k == 'workload' and path != ('config',):
                    continue
                if (depth > 1 or (depth == 1 and top not in ('roofline', 'config', 'cpu_baseline'))) and k in derivable and top != 'cpu_baseline':
                    continue
                out[k] = walk(v, path + (k,))
            return out
        if isinstance(o, list):
            return [walk(v, path + ('[]',)) for v in o]
        if isinstance(o, float) and depth > 1 and not (path[0] == 'roofline' and depth == 2) and path[-1] not in ('value', 'ms_per_step'):
            return float(f'{o:.6g}')
        return o

    return walk(line, ())


PROFILER_ENV_PREFIXES = ('ROCP', 'ROCTRACER', 'HSA_TOOLS', 'ROCTX')


def under_profiler(env=None):
    """True when a rocprofv3 / rocprofiler tool library is injected into this process (its children would inherit it)."""
    env = os.environ if env is None else env
    if any('rocprof' in env.get(k, '').lower() for k in ('LD_PRELOAD', 'HSA_TOOLS_LIB')):
        return True
    return any(k.startswith(('ROCPROFILER_', 'ROCPROF_', 'ROCP_')) for k in env)


def scrubbed_env(env=None):
    """Environment for helper children (rocm-smi): no preloaded profiler library, no profiler variables -- a child that initialises the GPU
    through an inherited tool library and then execs (a `#!/usr/bin/env` script does) is the hop this pool forbids."""
    env = dict(os.environ if env is None else env)
    for k in list(env):
        if k.startswith(PROFILER_ENV_PREFIXES):
            env.pop(k)
    if 'LD_PRELOAD' in env:                                       # only the profiler's entries go; whatever else the host preloads stays
        keep = [e for e in env['LD_PRELOAD'].replace(' ', ':').split(':') if e and 'rocprof' not in e.lower()]
        if keep:
            env['LD_PRELOAD'] = ':'.join(keep)
        else:
            env.pop('LD_PRELOAD')
    return env


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks as FRESH children under
    torch.distributed.run (the driver's own command shape) before this process has imported torch or touched the GPU, relay rank 0's one
    JSON line on stdout, pass everything else to stderr and return the launcher's exit code.  A rendezvous port that was free when probed can
    be taken before rank 0 binds it: that failure, and only that one, is retried on another port."""
    import socket
    import subprocess
    import threading
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'GROUP_RANK', 'LOCAL_WORLD_SIZE', 'ROLE_RANK', 'TORCHELASTIC_RUN_ID'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')            # dmabuf IPC only on this pool (RCCL across processes)
    rc = 1
    for attempt in range(4):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1', '--master-port', str(port),
               os.path.abspath(__file__)] + list(argv)
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, errors='replace')
        tail, json_lines = [], []

        def pump_err(pipe=proc.stderr, tail=tail):
            for ln in pipe:
                sys.stderr.write(ln)
                tail.append(ln)
                del tail[:-200]

        th = threading.Thread(target=pump_err, daemon=True)
        th.start()
        for ln in proc.stdout:
            if ln.startswith('{'):
                json_lines.append(ln)
            else:
                sys.stderr.write(ln)
        rc = proc.wait()
        th.join(timeout=10)
        if rc != 0 and not json_lines and attempt < 3 and any('ddress already in use' in ln or 'EADDRINUSE' in ln for ln in tail):
            sys.stderr.write(f'bench.py: rendezvous port {port} was taken, retrying on another one\n')
            continue
        for ln in json_lines:
            sys.stdout.write(ln)
        sys.stdout.flush()
        if rc == 0 and len(json_lines) != 1:
            sys.stderr.write(f'bench.py: expected one JSON line from rank 0, got {len(json_lines)}\n')
            return 1
        return rc
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', type=int, default=2, choices=[2, 3, 4, 5],
                    help='BASELINE.json config: 2 (headline, 65 536 trials), 3 (mixture + annealing, 262 144), 4 (config-2 inputs, 1 048 576 trials in total over all ranks), 5 (16-feature / 7-DoF stress)')
    ap.add_argument('--scaling', default=None, choices=['strong', 'weak'],
                    help="weak (default): every GPU runs the config's trial count, global trial numbering; strong: the config's trial count is the TOTAL, sharded over the "
                         "ranks (north_star's 65 536-trial series; the default for --config 4, which names a total).  A weak config-2 run on N > 1 ranks also times "
                         "the strong series as the side object multi_gpu.strong_series")
    ap.add_argument('--hold', action='store_true', help='config 3: hold outliers for 10 steps (noise.hold)')
    ap.add_argument('--trials', type=int, default=0, help="override the config's trial count (total for strong, per GPU for weak)")
    ap.add_argument('--lanes', type=int, default=0, help='lanes per filter (0 = library default)')
    ap.add_argument('--no-replay', action='store_true', help='skip the replay-mode (estimator kernel) side measurement')
    ap.add_argument('--force-dist', action='store_true', help='run the multi-rank code path (process group, barrier, stats gather) even with one rank: RCCL smoke test on a 1-GPU box')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-side', action='store_true', help='skip the side measurements of the default run (configs 3 / 3-hold / 5, other estimators, end-to-end sweep)')
    ap.add_argument('--e2e', action='store_true', help='only the headline and the end-to-end sweep side object')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end sweep side object (counter passes: its 49 launches of the headline kernel on other noise would be averaged in)')
    ap.add_argument('--backend', default=None, choices=['nccl', 'gloo'], help='torch.distributed backend (default: nccl = RCCL when every rank has a GPU of its own, gloo when ranks must share cards -- a 1-GPU box)')
    ap.add_argument('--latency', action='store_true', help='UVS_OPT_LATENCY on the timed launch: plain four-lane kernels for shards that do not fill the chip (a few % faster than the default small-batch kernels; results differ from the default mapping in the last bits)')
    ap.add_argument('--no-strong-series', action='store_true', help='N > 1, weak: skip the strong-series side measurement')
    ap.add_argument('--no-power', action='store_true', help='skip the power / clock samples under load (about 3 s of extra launches)')
    ap.add_argument('--stats-only', action='store_true', help='do not write the per-step X / err / q streams (separate line, B = noise read only)')
    ap.add_argument('--host-noise', action='store_true', help='generate the noise streams with numpy on the host (default: HIP generator)')
    ap.add_argument('--layout', default=None, choices=['kct', 'ktc', 'tkc'], help='physical layout of the per-step streams (default: trial-fastest kct; config 5: per-trial records ktc)')
    args = ap.parse_args()
    if under_profiler():                                          # rocprofv3 pass: no helper children from a GPU-initialised process, no 49 extra launches in the counters
        args.no_power = args.no_e2e = True

    if args.layout is None:
        args.layout = 'ktc' if args.config == 5 else 'kct'    # 8 trials per wavefront at (32,7): only per-trial records give contiguous stores
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:         # plain `python bench.py --gpus N`: this process only starts the ranks
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: either run plain `python bench.py --gpus N` (it starts its own ranks) or '
                 f'`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`')
    dist_on = world > 1 or args.force_dist
    scaling = args.scaling or ('strong' if args.config == 4 else 'weak')   # config 4 names a total; every other config is a per-GPU size
    assert not (args.config == 4 and scaling == 'weak'), 'config 4 names a TOTAL (1 048 576 trials over all ranks): it has no weak form'

    import uvs_amd
    from uvs_amd import batch, dist, engine

    cfg = config2()
    size = args.trials or CONFIG_TRIALS[args.config]
    trials_total = size * world if scaling == 'weak' else size
    assert trials_total >= world, 'fewer trials than ranks'
    cell = ALPHA
    if args.config == 3:                                          # BASELINE config 3 / BASELINE.md table
        cfg['noise'].update(type='GAUSSIAN_MIXTURE', noise_params={'std': 1.0, 'mean': 50.0, 'rho': 0.1}, hold=bool(args.hold), hold_time=0.5)
        cfg['estimator']['estimator_params']['annealing'] = True
        cell = 0.1
    cfg['experiments']['epoch'] = trials_total
    plan = batch.plan_trials(cfg, cells=[cell])                   # global enumeration: trial t -> seed 123456 + t, jitter draw t (main.py:121-139)
    lo, hi = dist.shard_range(len(plan), rank, world)
    t_log = engine.loop_clock(0.05, 15)
    K = len(t_log)
    headline_shape = args.config == 2 and hi - lo == TRIALS_PER_GPU and args.layout == 'kct' and args.lanes in (0, 2) and not args.stats_only

    # ---- everything that forks happens before the GPU is touched
    cores = available_cores()
    workers = max(1, cores // max(1, world))
    cpu = None
    assert not (args.host_noise and args.config not in (2, 4)), '--host-noise is wired for the config-2 inputs only'
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 2:
        cpu = cpu_baseline(plan.q_start)
    noise_host, gen_s = None, 0.0
    if args.host_noise:
        t0 = time.perf_counter()
        noise_host = host_noise(plan.seed[lo:hi], K, workers)
        gen_s = time.perf_counter() - t0

    import torch
    n_dev = max(1, torch.cuda.device_count())
    backend_note = None
    if args.backend is None:                                      # RCCL refuses two ranks on one card: ranks that must share cards gather over gloo
        args.backend = 'nccl' if world <= n_dev else 'gloo'
        if args.backend == 'gloo':
            backend_note = f'{world} ranks on {n_dev} visible GPU(s): ranks share cards, the statistics gather runs over gloo (host) instead of RCCL -- a functional run, not a scaling measurement'
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if dist_on:
        import torch.distributed as td
        if args.backend == 'nccl':
            td.init_process_group('nccl', device_id=dev)         # RCCL over xGMI
        else:
            td.init_process_group('gloo')
    ranks_seen = td.get_world_size() if dist_on else 1
    uvs_amd.lib()

    t0 = time.perf_counter()
    noise = None
    if args.config != 5:
        if args.host_noise:
            noise = torch.as_tensor(noise_host, device=dev)       # PCIe upload, outside the timed region
        else:                                                     # NoiseProfiler-compatible streams generated on the GPU
            noise = batch.device_noise(cfg, plan, lo, hi, K, dev, share=False)          # first call: + one-time table upload / torch warm-up
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            noise = batch.device_noise(cfg, plan, lo, hi, K, dev, share=False)          # what one more sweep cell costs
            torch.cuda.synchronize()
            gen_s = time.perf_counter() - t1
        if args.layout != 'kct':                                  # generated as [step][comp][trial]
            noise = noise.permute({'ktc': (0, 2, 1), 'tkc': (2, 0, 1)}[args.layout]).contiguous()
        q0 = torch.as_tensor(plan.q_start[lo:hi].copy(), device=dev)
    torch.cuda.synchronize()
    h2d_s = time.perf_counter() - t0
    del noise_host

    p = cfg['estimator']['estimator_params']
    M, N, x0 = 8, 6, None
    if args.config == 5:                                          # synthetic 16-feature / 7-DoF stress shape on the consistent linear plant
        M, N = 32, 7
        lin = uvs_amd.LinearPlant.random(M, N, seed=2)
        rng = np.random.default_rng(5)
        q_goal = lin.q0 + rng.uniform(-0.3, 0.3, N)
        desired = lin.features(q_goal)
        gq = np.random.default_rng(12345)
        q0 = torch.as_tensor(q_goal + gq.uniform(-0.15, 0.15, (len(plan), N))[lo:hi], device=dev)
        x0 = torch.as_tensor(np.tile((lin.J * (1 + 0.1 * rng.normal(size=lin.J.shape))).ravel(), (hi - lo, 1)), device=dev)
        noise = uvs_amd.noise_device.generate(uvs_amd.NoiseType.ALPHA_STABLE, dict(alpha=ALPHA, beta=0, gamma=1, delta=0), plan.seed[lo:hi], M, K, layout=args.layout, device=dev)
        fp = engine.make_params(M, N, 'GMCKF', p['kernel_bw'], p['annealing'], 0.05, 15, 0.2, desired, False, args.lanes)
        plant = lin.to_struct(dev)
    else:
        fp = engine.make_params(8, 6, 'GMCKF', p['kernel_bw'], p['annealing'], 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, args.lanes)
        plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    if args.latency:
        fp.reserved |= 2                                          # UVS_OPT_LATENCY
    T = hi - lo
    # output buffers are allocated once and reused by every step (engine.closed_loop allocates; here we pre-allocate by hand)
    bufs = {k: engine.alloc_stream(T, K, c, args.layout, dev, zero=True) for k, c in (('x', M * N), ('err', M), ('q', N))}
    stats = torch.zeros((T, 3), dtype=torch.float64, device=dev)
    status = torch.zeros(T, dtype=torch.int32, device=dev)
    k_done = torch.zeros(T, dtype=torch.int32, device=dev)
    import ctypes as C
    flat = lambda t: uvs_amd._lib.View(t.data_ptr(), t.stride(0), 0, t.stride(1))     # noqa: E731
    NV = uvs_amd._lib.NULL_VIEW

    def launch():
        rc = engine.launch_closed_loop(fp, plant, T, flat(q0), engine.stream_view(noise, args.layout), NV if x0 is None else flat(x0),
            NV if args.stats_only else engine.stream_view(bufs['x'], args.layout),
            NV if args.stats_only else engine.stream_view(bufs['err'], args.layout), NV if args.stats_only else engine.stream_view(bufs['q'], args.layout), NV, NV, stats.data_ptr(), status.data_ptr(),
            k_done.data_ptr(), NV, NV, device=dev)
        uvs_amd._lib.check(rc)

    def gather():
        rows = dist.pack_rows(stats, status)
        return dist.gather_trial_rows(rows if args.backend == 'nccl' else rows.cpu(), len(plan))

    def barrier():
        if dist_on:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        launch()
        if dist_on:
            gather()
    barrier()
    kernel_ev, gather_host_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()                                               # HIP events on the stream the kernel is launched on
        launch()
        e1.record()
        if dist_on:
            g0 = time.perf_counter()
            gathered = gather()
            gather_host_ms.append((time.perf_counter() - g0) * 1e3)   # gloo: host-side; nccl: enqueue time only, see the events
        e2.record()
        kernel_ev.append((e0, e1, e2))
    barrier()
    wall = time.perf_counter() - t0
    red_dev = dev if args.backend == 'nccl' else torch.device('cpu')
    kernel_ms = [a.elapsed_time(b) for a, b, _ in kernel_ev]
    gather_ms = [b.elapsed_time(c) for _, b, c in kernel_ev] if args.backend == 'nccl' else gather_host_ms
    avg_ms = float(np.mean(kernel_ms))
    rank_ms = {'max': avg_ms, 'min': avg_ms}
    gather_avg = float(np.mean(gather_ms)) if dist_on else 0.0
    if dist_on:
        w = torch.tensor([wall, avg_ms, -avg_ms, gather_avg], dtype=torch.float64, device=red_dev)
        td.all_reduce(w, op=td.ReduceOp.MAX)
        wall, rank_ms, gather_avg = float(w[0]), {'max': float(w[1]), 'min': -float(w[2])}, float(w[3])
        assert gathered.shape == (len(plan), 4)

    updates_per_launch = int(k_done.sum().item())                 # FAIL trials stop early; count what was actually computed
    failed_local = int((status != 0).sum().item())                # of THIS workload: the side measurements below re-use the status buffer
    if dist_on:
        u = torch.tensor([updates_per_launch, failed_local], dtype=torch.int64, device=red_dev)
        td.all_reduce(u)
        total_updates, failed_total = int(u[0]), int(u[1])
    else:
        total_updates, failed_total = updates_per_launch, failed_local
    value = total_updates * args.steps / wall

    strong_side = None
    if dist_on and world > 1 and scaling == 'weak' and args.config == 2 and not args.stats_only and not args.no_strong_series:
        # north_star's series -- ONE 65 536-trial batch over the N ranks -- timed in the same job, with the same barriers and the same gather,
        # right after the weak run (side object: never `value`).  Rank r owns dist.shard_range(65 536, r, N) of the global enumeration.
        cfg_s = config2()
        cfg_s['experiments']['epoch'] = args.trials or TRIALS_PER_GPU
        plan_s = batch.plan_trials(cfg_s, cells=[ALPHA])
        lo_s, hi_s = dist.shard_range(len(plan_s), rank, world)
        Ts = hi_s - lo_s
        noise_s = batch.device_noise(cfg_s, plan_s, lo_s, hi_s, K, dev, share=False)
        q0_s = torch.as_tensor(plan_s.q_start[lo_s:hi_s].copy(), device=dev)
        bufs_s = {k: engine.alloc_stream(Ts, K, c, 'kct', dev, zero=True) for k, c in (('x', M * N), ('err', M), ('q', N))}
        stats_s = torch.zeros((Ts, 3), dtype=torch.float64, device=dev)
        status_s = torch.zeros(Ts, dtype=torch.int32, device=dev)
        k_done_s = torch.zeros(Ts, dtype=torch.int32, device=dev)
        fp_s = engine.make_params(8, 6, 'GMCKF', p['kernel_bw'], p['annealing'], 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, args.lanes)
        strong_runs = {}

        def launch_s():
            rc = engine.launch_closed_loop(fp_s, plant, Ts, flat(q0_s), engine.stream_view(noise_s, 'kct'), NV, engine.stream_view(bufs_s['x'], 'kct'),
                engine.stream_view(bufs_s['err'], 'kct'), engine.stream_view(bufs_s['q'], 'kct'), NV, NV, stats_s.data_ptr(), status_s.data_ptr(),
                k_done_s.data_ptr(), NV, NV, device=dev)
            uvs_amd._lib.check(rc)

        def gather_s():
            rows = dist.pack_rows(stats_s, status_s)
            return dist.gather_trial_rows(rows if args.backend == 'nccl' else rows.cpu(), len(plan_s))

        for mode, bits in (('default_mapping', 0), ('latency_mapping', 2)):
            fp_s.reserved = bits
            for _ in range(max(1, args.warmup)):
                launch_s()
                gather_s()
            barrier()
            ev_s = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch_s()
                e1.record()
                gathered_s = gather_s()
                ev_s.append((e0, e1))
            barrier()
            wall_s = time.perf_counter() - t0
            k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_s]))
            w = torch.tensor([wall_s, k_ms, -k_ms], dtype=torch.float64, device=red_dev)
            td.all_reduce(w, op=td.ReduceOp.MAX)
            u = torch.tensor([int(k_done_s.sum().item()), int((status_s != 0).sum().item())], dtype=torch.int64, device=red_dev)
            td.all_reduce(u)
            assert gathered_s.shape == (len(plan_s), 4)
            strong_runs[mode] = {'ms_per_step': float(w[0]) / args.steps * 1e3, 'value': int(u[0]) * args.steps / float(w[0]), 'unit': 'updates/s',
                                 'kernel_ms_avg_over_ranks': {'max': float(w[1]), 'min': -float(w[2])}, 'failed_trials': int(u[1]),
                                 'lanes_per_filter': int(uvs_amd.lib().uvs_rmckf_closed_loop_lanes(C.byref(fp_s), C.byref(plant), Ts))}
        strong_side = {'scaling': 'strong', 'trials_total': len(plan_s), 'trials_rank0': Ts, 'n_gpus': world, 'steps': args.steps, **strong_runs['default_mapping'],
                       'latency_mapping': strong_runs['latency_mapping'],
                       'note': "north_star's series (one 65 536-trial batch sharded over the ranks + the same all-gather), timed with the same barriers right after the weak "
                               'run of this job; a side object, never `value`.  A trial is a 299-step serial chain: shards below one round of wavefronts leave SIMDs idle.  '
                               'The default mapping runs shards up to 16 384 trials on four lanes per filter with the two-lane arithmetic (bit-identical to the 1-GPU sweep); '
                               'latency_mapping = the same with UVS_OPT_LATENCY (plain four-lane kernels: a few % faster, last-bit differences)'}
        del bufs_s, noise_s

    if rank == 0:
        b_alg = 8 * (2 * M + N + M * N)                           # 560 B / update at (8,6): noise in, err + X + q out
        if args.stats_only:
            b_alg = 8 * M                                         # only the noise stream is read; statistics are 24 B per trial
        achieved = updates_per_launch * b_alg / (avg_ms * 1e-3) / 1e9
        traffic, tr_src, valu = None, None, None
        tr_path = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
        tr = json.load(open(tr_path)) if os.path.exists(tr_path) else None
        if tr is not None and tr.get('library_version') != uvs_amd.lib().uvs_version().decode():
            # counter figures of other kernels than the ones that just ran are not reported (tests/test_host_logic.py fails on the same condition)
            tr_src = f"profiles/traffic_latest.json is stale (counted on {tr.get('library_version')!r}, this library is {uvs_amd.lib().uvs_version().decode()!r}): not used"
            tr = None
        if tr is not None and headline_shape:
            traffic = tr.get('hbm_bytes_per_launch')                # rocprofv3 PMC, measured on exactly this launch shape -- a committed
            tr_src = f"profiles/traffic_latest.json (round {tr.get('round')}, {tr.get('source')}): rocprofv3 PMC passes of this launch shape, not a measurement of this run"
            if tr.get('valu_wave_instr_per_launch'):               # what binds: instruction issue ("report fp64 FLOP/s alongside GB/s")
                wi, fl = tr['valu_wave_instr_per_launch'], tr.get('fp64_flop_per_launch')
                valu = {'wave_instr_per_launch': wi, 'wave_instr_per_s': wi / (avg_ms * 1e-3), 'peak': VALU_PEAK_WAVE_INSTR_PER_S,
                        'peak_note': '1024 SIMDs x one wave64 VALU instruction per 4 cycles x 2.4 GHz peak clock (the part runs this kernel at 1.9-2.1 GHz)',
                        'frac': wi / (avg_ms * 1e-3) / VALU_PEAK_WAVE_INSTR_PER_S,
                        'fp64_flops': (fl / (avg_ms * 1e-3)) if fl else None, 'fp64_peak_flops': FP64_VECTOR_PEAK_FLOPS,
                        'fp64_frac': (fl / (avg_ms * 1e-3) / FP64_VECTOR_PEAK_FLOPS) if fl else None,
                        'source': 'instruction counts per launch from the committed PMC passes (SQ_INSTS_VALU, SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64), divided by this run\'s kernel time'}
        side_ok = world == 1 and headline_shape and not args.no_side
        replay = None
        if world == 1 and args.config == 2 and not args.no_replay and not args.e2e and not args.stats_only and args.layout == 'kct' and args.lanes in (0, 2):
            replay = replay_side_measurement(torch, engine, uvs_amd, fp, plant, q0, bufs['x'], bufs['err'], T, K, M, N)
        others = None
        if side_ok and not args.e2e:
            # the reference's other three estimators on the headline workload: same inputs, same streams logged;
            # MCKF once more on alpha = 1.0 noise (the reference's first sweep cell), where its fixed-point iteration and its FAIL path are live
            others = {}
            fp_head, noise_head = fp, noise
            cfg1 = config2()
            cfg1['experiments']['epoch'] = trials_total
            cfg1['noise']['noise_params']['alpha'] = 1.0
            plan1 = batch.plan_trials(cfg1, cells=[1.0])
            for key, meth, alpha in (('KF', 'KF', ALPHA), ('IMCCKF', 'IMCCKF', ALPHA), ('MCKF', 'MCKF', ALPHA), ('MCKF_alpha1', 'MCKF', 1.0)):
                fp = engine.make_params(M, N, meth, fp_head.kernel_bw, bool(fp_head.annealing), fp_head.dt, fp_head.dt * fp_head.k_max, fp_head.gain,
                                        list(fp_head.desired)[:M], bool(fp_head.initial_guess), args.lanes)
                if alpha != ALPHA:
                    noise = batch.device_noise(cfg1, plan1, lo, hi, K, dev, share=False)
                ms = []
                for i in range(2 + 5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch()
                    e1.record()
                    torch.cuda.synchronize()
                    if i >= 2:
                        ms.append(e0.elapsed_time(e1))
                upd = int(k_done.sum().item())
                avg = float(np.mean(ms))
                rec_ms = None
                if meth in ('KF', 'IMCCKF') and args.layout == 'kct' and T % 16 == 0:
                    # what shipped in round 5 for these two: X written as per-trial records ([step][trial][component], engine.closed_loop(x_layout='ktc')),
                    # the narrow streams unchanged -- the same buffer viewed as records, same bits
                    x_rec = (bufs['x']._base if bufs['x']._base is not None else bufs['x']).reshape(-1)[:K * T * M * N].view(K, T, M * N)
                    rms = []
                    for i in range(2 + 5):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        uvs_amd._lib.check(engine.launch_closed_loop(fp, plant, T, flat(q0), engine.stream_view(noise, 'kct'), NV, engine.stream_view(x_rec, 'ktc'),
                                                                     engine.stream_view(bufs['err'], 'kct'), engine.stream_view(bufs['q'], 'kct'), NV, NV, stats.data_ptr(),
                                                                     status.data_ptr(), k_done.data_ptr(), NV, NV, device=dev))
                        e1.record()
                        torch.cuda.synchronize()
                        if i >= 2:
                            rms.append(e0.elapsed_time(e1))
                    rec_ms = float(np.mean(rms))
                others[key] = {'avg_kernel_ms': avg, 'updates_per_s': upd / (avg * 1e-3), 'achieved': upd * b_alg / (avg * 1e-3) / 1e9, 'unit': 'GB/s',
                               'frac': upd * b_alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 'launches_timed': 5, 'alpha': alpha, 'updates_per_launch': upd,
                               'failed_trials': int((status != 0).sum().item()),
                               'work_items_per_trial': int(uvs_amd.lib().uvs_rmckf_closed_loop_segments(C.byref(fp), C.byref(plant), T)),
                               'hand_over_fallbacks': engine.hand_over_fallbacks(fp, plant, T, dev)}
                if rec_ms is not None:
                    others[key]['x_records'] = {'avg_kernel_ms': rec_ms, 'frac': upd * b_alg / (rec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'x_layout': 'ktc'}
                noise = noise_head
            fp = fp_head
            # UVS_OPT_STRICT_PINV on the headline workload: every control-law solve certified in-kernel (round 6) against the default mode's watches
            fp_strict = type(fp).from_buffer_copy(fp)
            fp_strict.reserved |= 1
            fp = fp_strict
            ms = []
            for i in range(0 if under_profiler() else 2 + 5):    # (not under rocprofv3: the same instantiation and grid as the headline would mix into its per-kernel counters)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch()
                e1.record()
                torch.cuda.synchronize()
                if i >= 2:
                    ms.append(e0.elapsed_time(e1))
            fp = fp_head
            if ms:
                others['strict_pinv'] = {'avg_kernel_ms': float(np.mean(ms)), 'x_default': float(np.mean(ms)) / avg_ms, 'failed_trials': int((status != 0).sum().item())}
        shard_model = None
        if side_ok and not args.e2e:
            # What ONE GPU does with the shard a rank of north_star's strong series would own (65 536 / N trials): kernel time per launch on the
            # first T/N trials of this run's inputs (strided views of the same buffers), default mapping and UVS_OPT_LATENCY.  One-GPU shard
            # timings -- what the series starts from on every rank -- NOT a scaling curve: no second GPU, no gather.
            shard_model = {'note': 'one-GPU shard timing, not a scaling curve: kernel ms per launch on the first 65 536 / N trials of the headline inputs; '
                                   'implied_speedup = (N = 1 kernel time) / (shard kernel time), an upper bound of what N ranks can reach before the gather.  '
                                   'two_lanes = the headline mapping forced (lanes_per_filter 2); default_mapping = the library choice (up to 16 384 trials four lanes per '
                                   'filter with the two-lane arithmetic: bit-identical results); latency_mapping = UVS_OPT_LATENCY (plain four-lane kernels, last-bit differences)',
                           'n1_kernel_ms': avg_ms, 'shards': {}}
            for n_ranks in (2, 4, 8):
                Ts = T // n_ranks
                entry = {'trials': Ts}
                for mode, lanes_m, bits in (('two_lanes', 2, 0), ('default_mapping', 0, 0), ('latency_mapping', 0, 2)):
                    fp_m = type(fp).from_buffer_copy(fp)
                    fp_m.lanes_per_filter, fp_m.reserved = lanes_m, bits
                    ms = []
                    for i in range(3 + 8):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        rc = engine.launch_closed_loop(fp_m, plant, Ts, flat(q0[:Ts]), engine.stream_view(noise[:, :, :Ts], 'kct'), NV, engine.stream_view(bufs['x'][:, :, :Ts], 'kct'),
                                                       engine.stream_view(bufs['err'][:, :, :Ts], 'kct'), engine.stream_view(bufs['q'][:, :, :Ts], 'kct'), NV, NV, stats.data_ptr(),
                                                       status.data_ptr(), k_done.data_ptr(), NV, NV, device=dev)
                        uvs_amd._lib.check(rc)
                        e1.record()
                        torch.cuda.synchronize()
                        if i >= 3:
                            ms.append(e0.elapsed_time(e1))
                    entry[mode] = {'kernel_ms': float(np.mean(ms)), 'lanes_per_filter': int(uvs_amd.lib().uvs_rmckf_closed_loop_lanes(C.byref(fp_m), C.byref(plant), Ts)),
                                   'implied_speedup': avg_ms / float(np.mean(ms))}
                shard_model['shards'][f'N={n_ranks}'] = entry
            # the reference's shipped estimator (config.json: MCKF) on the same shards: two lanes per filter (round 4's only mapping) against the library's
            # choice (round 5: four lanes with the two-lane bits up to 16 384 trials, every fixed-point pass in-kernel)
            shard_model['mckf'] = {}
            for n_ranks in (4, 8):
                Ts = T // n_ranks
                entry = {'trials': Ts}
                for mode, lanes_m in (('two_lanes', 2), ('default_mapping', 0)):
                    fp_m = engine.make_params(M, N, 'MCKF', fp.kernel_bw, bool(fp.annealing), fp.dt, fp.dt * fp.k_max, fp.gain, list(fp.desired)[:M], bool(fp.initial_guess), lanes_m)
                    ms = []
                    for i in range(2 + 5):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        rc = engine.launch_closed_loop(fp_m, plant, Ts, flat(q0[:Ts]), engine.stream_view(noise[:, :, :Ts], 'kct'), NV, engine.stream_view(bufs['x'][:, :, :Ts], 'kct'),
                                                       engine.stream_view(bufs['err'][:, :, :Ts], 'kct'), engine.stream_view(bufs['q'][:, :, :Ts], 'kct'), NV, NV, stats.data_ptr(),
                                                       status.data_ptr(), k_done.data_ptr(), NV, NV, device=dev)
                        uvs_amd._lib.check(rc)
                        e1.record()
                        torch.cuda.synchronize()
                        if i >= 2:
                            ms.append(e0.elapsed_time(e1))
                    entry[mode] = {'kernel_ms': float(np.mean(ms)), 'lanes_per_filter': int(uvs_amd.lib().uvs_rmckf_closed_loop_lanes(C.byref(fp_m), C.byref(plant), Ts))}
                shard_model['mckf'][f'N={n_ranks}'] = entry
        power = None
        if world == 1 and not args.no_power:                       # after the short side measurements above (seconds of sustained load change what follows)
            power = power_under_load(torch, launch, torch.cuda.current_device())
        if power and valu:                                         # the issue peak at the clock the part actually sustains under this kernel
            clk = float(np.mean(power['sclk_mhz'])) * 1e6
            valu['sustained_clock_hz'] = clk
            valu['frac_at_sustained_clock'] = valu['wave_instr_per_s'] / (1024 * clk / 4)
        side = {}
        if side_ok:
            for key in list(bufs):
                bufs[key] = None                                   # 9.7 GB of config-2 streams: make room for 44 GB (config 3) / 41 GB (config 5)
            del noise
            torch.cuda.empty_cache()
            side['drop_in'] = drop_in_step_latency(torch, uvs_amd, engine, dev)
            if not args.no_e2e:
                side['e2e'] = e2e_sweep(torch, uvs_amd, engine, batch, dev)
            if not args.e2e:
                side['config3'] = side_config(3, torch, uvs_amd, engine, batch, dev)
                side['config3_hold'] = side_config(3, torch, uvs_amd, engine, batch, dev, hold=True)
                side['config5'] = side_config(5, torch, uvs_amd, engine, batch, dev)
        series = {'strong': f'strong scaling: {trials_total} trials in total, sharded contiguously over {world} rank(s)' +
                            (" -- north_star's series (a 65 536-trial batch at 1, 2, 4 and 8 MI355X); a trial is a 299-step serial chain, so by kernel time this "
                             "series tops out at <= 3.3x at N = 8 (8 192-trial shard 0.93 ms vs 3.0 ms); the weak series / config 4 is the scaling series" if args.config == 2 and trials_total == TRIALS_PER_GPU else '') +
                            (' -- BASELINE config 4' if args.config == 4 else ''),
                  'weak': f'weak scaling: {size} trials on every one of {world} rank(s), global trial numbering (the scaling series; the strong 65 536-trial series of north_star is '
                          'bounded at <= 3.3x at N = 8 by kernel time -- a trial is a 299-step serial chain -- and is reported as multi_gpu.strong_series)'}[scaling]
        line = {
            'metric': 'RMCKF updates/s (4-feat, 6-DoF) over MC batch', 'value': value, 'unit': 'updates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': {2: 'BASELINE config 2: 4-feature UR10 closed loop, GMCKF(RMCKF) sigma=10, alpha-stable noise alpha=1.5, ',
                                    3: f'BASELINE config 3: 4-feature UR10 closed loop, GMCKF(RMCKF) annealed sigma, Gaussian mixture rho=0.1 mean=50 hold={bool(args.hold)}, ',
                                    4: 'BASELINE config 4: config-2 inputs (4-feature UR10 closed loop, GMCKF(RMCKF) sigma=10, alpha-stable alpha=1.5), 1 048 576 trials sharded over the ranks + all-gather of [ISE, IAE, ITAE, status], ',
                                    5: 'BASELINE config 5: synthetic 16-feature / 7-DoF (m=32, n=7) linear plant, GMCKF(RMCKF) sigma=10, alpha-stable alpha=1.5, '}[args.config] +
                                   f'{trials_total} trials in total ({T} on rank 0) x {K} updates, ' + ('statistics only (no per-step streams)' if args.stats_only else 'X+err+q logged per step'),
                       'series': series, 'trials_total': trials_total, 'trials_rank0': T, 'trials_per_gpu': T, 'updates_per_trial': K, 'ranks_seen': ranks_seen,
                       'lanes_per_filter': int(uvs_amd.lib().uvs_rmckf_closed_loop_lanes(C.byref(fp), C.byref(plant), T)), 'latency_option': bool(args.latency), 'layout': args.layout, 'failed_trials': failed_total},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': tr_src, 'kernel': 'closed_loop_tuned_kernel<8,6,2,GMCKF,DH(axis-aligned UR10 table),2,true>' if (args.config != 5 and args.lanes in (0, 2)) else 'closed_loop kernel, see lanes_per_filter', 'avg_kernel_ms': avg_ms,
                         'algorithmic_bytes_per_update': b_alg, 'updates_per_launch': updates_per_launch,
                         'binds': 'HBM is the roofline BASELINE.json prescribes; the counters say the kernel is bound by VALU issue at one wavefront per SIMD (see `valu`), at the clock the package power cap leaves it (see `power`)',
                         'valu': valu, 'power': power},
            'multi_gpu': {'kernel_ms_avg_over_ranks': rank_ms, 'gather_ms': gather_avg, 'gather_inside_timed_region': bool(dist_on), 'backend': args.backend if dist_on else None, 'backend_note': backend_note,
                          'gather_note': 'all_gather of per-trial [ISE, IAE, ITAE, status] rows (32 B/trial); nccl: HIP events on the launch stream, gloo: host clock',
                          'strong_series': strong_side, 'shard_model': shard_model},
            'cpu_baseline': cpu,
            'replay': replay,
            'config3': side.get('config3'), 'config3_hold': side.get('config3_hold'), 'config5': side.get('config5'), 'other_estimators': others, 'e2e': side.get('e2e'), 'drop_in': side.get('drop_in'),
            'setup': {'noise': 'host numpy' if args.host_noise else 'device (uvs_noise_generate_f64)', 'noise_gen_s': gen_s,
                      'noise_gen_workers': workers if args.host_noise else 0, 'h2d_s': h2d_s,
                      'h2d_inclusive_updates_per_s': total_updates / (wall / args.steps + h2d_s) if (world == 1 and args.host_noise) else None},
        }
        full_path = os.environ.get('UVS_BENCH_FULL_JSON')
        if full_path:                                              # the unabridged object, notes included (profiles/rNN/bench_full.json is written this way)
            with open(full_path, 'w') as fh:
                json.dump(line, fh, indent=1)
        print(json.dumps(compact_line(line), separators=(',', ':')))
    if dist_on:
        td.destroy_process_group()


if __name__ == '__main__':
    main()

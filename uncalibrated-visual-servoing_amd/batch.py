"""Monte-Carlo driver: what main.py of the reference does, as one HIP grid per sweep instead of sequential trials.

Keeps the reference's ``config.json`` schema (main.py:16-98), sweep cells (``linspace(0, .2, 12)`` of rho, or
``linspace(1, 2, 12)`` of alpha for ALPHA_STABLE, main.py:104-106), global trial numbering, per-trial noise seed
``seed + trial`` (main.py:137-139), the q_start jitter stream ``NoiseProfiler(2, UNIFORM, seed=experiment_seed)`` with
the ``2 (r - 1)`` formula (main.py:118, 132-134) and the 41-column results.csv (main.py:152-196; SURVEY.md says 43, the DataFrame has 4 + 6 + 6 + 8 + 8 + 8 + 1 = 41).
"""
import json
import os
from dataclasses import dataclass, field

import numpy as np

from . import dist, engine, noise_device
from .experiment import ExperimentStatus, Method, bandwidth_log
from .noise import NoiseProfiler, NoiseType, noise_batch
from .plant import SyntheticPlant

REQUIRED = {'experiments': ('dt', 't_max', 'epoch', 'ibvs_gain', 'q_start', 'desired_f', 'visualization', 'change_q_start', 'seed'),
            'estimator': ('method', 'estimator_params'),
            'noise': ('type', 'noise_params', 'hold', 'hold_time', 'seed')}


def load_config(path_or_dict):
    """Parse and validate a reference-format config (main.py:16-98); unknown method / noise names and missing keys raise."""
    cfg = path_or_dict if isinstance(path_or_dict, dict) else json.load(open(path_or_dict, 'r', encoding='utf-8'))
    if 'log_level' not in cfg:
        raise KeyError('log_level')
    for section, keys in REQUIRED.items():
        for key in keys:
            if key not in cfg[section]:
                raise KeyError(f'{section}.{key}')
    if cfg['estimator']['method'] not in Method.__members__:
        raise ValueError('Estimation method ' + cfg['estimator']['method'] + ' unknown.')
    if cfg['noise']['type'] not in NoiseType.__members__:
        raise ValueError('Noise type ' + cfg['noise']['type'] + ' unknown.')
    return cfg


def sweep_cells(noise_type):
    return np.linspace(1, 2, 12) if noise_type == NoiseType.ALPHA_STABLE else np.linspace(0, 0.2, 12)   # main.py:104-106


@dataclass
class TrialPlan:
    """Global enumeration of a sweep: trial t belongs to cell ``cell[t]`` with swept value ``value[t]``."""
    cell: np.ndarray
    value: np.ndarray
    seed: np.ndarray
    q_start: np.ndarray
    cells: np.ndarray = field(default=None)

    def __len__(self):
        return len(self.cell)


def plan_trials(cfg, cells=None, epoch=None):
    """Enumerate trials like main.py:121-139.  ``cells`` overrides the 12-point sweep (e.g. [1.5]); ``epoch`` the trials per cell."""
    ex, nz = cfg['experiments'], cfg['noise']
    noise_type = NoiseType[nz['type']]
    cells = sweep_cells(noise_type) if cells is None else np.asarray(cells, float)
    epoch = int(ex['epoch'] if epoch is None else epoch)
    T = len(cells) * epoch
    q0 = np.tile(np.asarray(ex['q_start'], float), (T, 1))
    if ex['change_q_start']:
        # two UNIFORM generators seeded seed and seed + 10 (noise.py:70); trial t consumes the t-th draw of each
        g0, g1 = (NoiseProfiler(2, NoiseType.UNIFORM, seed=ex['seed']).generators[i] for i in range(2))
        q0[:, 0] += 2 * (g0.random(T) - 1) * (np.pi / 18)           # main.py:133
        q0[:, 1] += 2 * (g1.random(T) - 1) * (np.pi / 9)            # main.py:134
    seed0 = nz['seed']
    if seed0 is not None:
        seeds = seed0 + np.arange(T)                                # main.py:137-139: seed, seed + 1, ... over the whole sweep
    else:
        # seed: null is legal in the reference: main.py:138 skips the increment and NoiseProfiler(seed=None) falls back to
        # default_rng() per generator (noise.py:61-70), i.e. every trial gets independent OS-entropy streams.  Draw one fresh
        # 62-bit seed per trial (so that seed + 10 j and 2 seed + i stay below 2^64) and keep them in the plan / npz sink:
        # the run is then reproducible after the fact, and no two trials share a noise realisation.
        seeds = (np.random.SeedSequence().generate_state(T, np.uint64) >> np.uint64(2)).astype(np.int64)
    return TrialPlan(cell=np.repeat(np.arange(len(cells)), epoch), value=np.repeat(cells, epoch), seed=seeds, q_start=q0, cells=cells)


def trial_noise(cfg, plan, lo, hi, steps, out):
    """Fill ``out`` (logical [trial][step][m] array-like covering trials lo..hi) with the reference's noise streams."""
    nz = cfg['noise']
    noise_type = NoiseType[nz['type']]
    m = len(cfg['experiments']['desired_f'])
    hold_cnt = int(nz['hold_time'] / cfg['experiments']['dt'])      # main.py:137
    key = 'alpha' if noise_type == NoiseType.ALPHA_STABLE else 'rho'
    for c in np.unique(plan.cell[lo:hi]):
        idx = np.nonzero(plan.cell[lo:hi] == c)[0]
        params = dict(nz['noise_params'])
        params[key] = float(plan.cells[c])                          # main.py:123-126
        block = noise_batch(noise_type, params, plan.seed[lo:hi][idx], m, steps, nz['hold'], hold_cnt)
        out[idx] = block
    return out


@dataclass
class BatchResult:
    plan: TrialPlan
    lo: int
    hi: int
    t: np.ndarray
    stats: object            # (T_local, 3) cuda tensor
    status: object           # (T_local,) int32 cuda tensor
    k_done: object
    streams: dict            # optional per-step tensors, trial-fastest [K][comp][T_local]
    noise: object
    seconds: float = 0.0


def device_noise(cfg, plan, lo, hi, steps, device='cuda', share=True):
    """Noise streams of trials lo..hi generated on the GPU (noise_device.py), [step][feature][trial].  ``share``: where the trials' streams
    alias (noise_device.shares_streams) generate each distinct stream once and return an overlapping view."""
    nz = cfg['noise']
    noise_type = NoiseType[nz['type']]
    m = len(cfg['experiments']['desired_f'])
    hold_cnt = int(nz['hold_time'] / cfg['experiments']['dt'])      # main.py:137
    key = 'alpha' if noise_type == NoiseType.ALPHA_STABLE else 'rho'
    torch = engine._torch()
    cells_here = np.unique(plan.cell[lo:hi])
    if share and len(cells_here) == 1 and noise_device.shares_streams(noise_type, nz['hold'], plan.seed[lo:hi]):
        # one cell, consecutive seeds, no hold: the T + 70 distinct streams once, read through an overlapping [step][feature][trial] view
        # (main.py:137-139 + noise.py:70; a shard's rank generates seeds lo ... hi + 70) -- bit-identical to the per-trial generation below
        params = dict(nz['noise_params'])
        params[key] = float(plan.cells[cells_here[0]])
        return noise_device.generate_shared(noise_type, params, int(plan.seed[lo]), hi - lo, m, steps, device=device)[1]
    out = engine.alloc_stream(hi - lo, steps, m, 'kct', device)
    seeds_dev = torch.as_tensor(np.ascontiguousarray(plan.seed[lo:hi], dtype=np.uint64).view(np.int64), device=device)   # one upload for the shard
    for c in np.unique(plan.cell[lo:hi]):
        idx = np.nonzero(plan.cell[lo:hi] == c)[0]                  # cells are contiguous runs of trials (main.py:121-127)
        params = dict(nz['noise_params'])
        params[key] = float(plan.cells[c])                          # main.py:123-126
        a, b = int(idx[0]), int(idx[-1]) + 1
        noise_device.generate(noise_type, params, seeds_dev[a:b], m, steps, nz['hold'], hold_cnt, 'kct', out=out[:, :, a:b], device=device)
    return out


def run_batch(cfg, plant=None, cells=None, epoch=None, rank=0, world=1, want=('err',), lanes=0, noise_tensor=None,
              device='cuda', noise_on_device=True, strict_pinv=False, latency=False):
    """Run this rank's shard of the sweep on its GPU.  Returns a BatchResult whose tensors stay on the device.
    ``noise_on_device``: generate the measurement noise with the HIP generator (default) or with numpy on the host.
    ``strict_pinv`` / ``latency``: the library's launch options UVS_OPT_STRICT_PINV (numpy's pinv on every control-law solve, slower) and
    UVS_OPT_LATENCY (four lanes per filter for shards that do not fill the chip: last-bit differences) -- not part of config.json, whose
    schema stays the reference's."""
    import torch
    cfg = load_config(cfg)
    ex, est = cfg['experiments'], cfg['estimator']
    method = Method[est['method']]
    if method == Method.ANALYTICAL:
        raise NotImplementedError('ANALYTICAL is not an estimator (and crashes in the reference: R is unbound, experiment.py:121)')
    plant = SyntheticPlant.ur10(ex['desired_f']) if plant is None else plant
    plan = plan_trials(cfg, cells, epoch)
    lo, hi = dist.shard_range(len(plan), rank, world)
    p = est['estimator_params']
    m, n = len(ex['desired_f']), plant.n_joints
    fp = engine.make_params(m, n, method.name, p.get('kernel_bw', 1.0), p.get('annealing', False), ex['dt'], ex['t_max'],
                            ex['ibvs_gain'], ex['desired_f'], p['initial_guess'], lanes, None, p.get('fpi_threshold', 0.1),
                            p.get('fpi_epoch_max', 1000))
    fp.reserved = (1 if strict_pinv else 0) | (2 if latency else 0)
    t_log = engine.loop_clock(ex['dt'], ex['t_max'])
    K, Tl = len(t_log), hi - lo
    dev = torch.device(device)
    with torch.cuda.device(dev if dev.index is not None else torch.cuda.current_device()):   # the engine launches on the CURRENT device's current stream
        if noise_tensor is None and noise_on_device:
            noise_tensor = device_noise(cfg, plan, lo, hi, K, dev)
        if noise_tensor is None:
            host = np.empty((K, m, Tl))
            trial_noise(cfg, plan, lo, hi, K, host.transpose(2, 0, 1))  # logical [trial][step][m] view of the trial-fastest buffer
            noise_tensor = torch.as_tensor(host, device=dev)
        q0 = torch.as_tensor(plan.q_start[lo:hi].copy(), device=dev)
        x0 = None
        if not p['initial_guess']:
            x0 = torch.as_tensor(np.asarray(p['x0'], float).reshape(1, m * n).repeat(Tl, 0), device=dev)
        out = engine.closed_loop(fp, plant.to_struct(), q0, noise_tensor, x0, want=want)
    start, stop = out['events']                                     # HIP events around the kernel launch (output allocation excluded)
    stop.synchronize()
    return BatchResult(plan, lo, hi, t_log, out['stats'], out['status'], out['k_done'],
                       {k: out[k] for k in ('x', 'err', 'q', 'f', 'dq') if out.get(k) is not None}, noise_tensor,
                       start.elapsed_time(stop) * 1e-3)


_COPY_STREAMS = {}


def sweep_pieces(plan, lo, hi, max_trials=None):
    """[(a, b, cell)]: this shard's trials [lo, hi) cut at the cell boundaries (main.py:121-127: a cell is a contiguous run of trials) and,
    when ``max_trials`` is given, into pieces of at most that many trials -- the launches of run_sweep, in trial order."""
    out, a = [], lo
    while a < hi:
        c = int(plan.cell[a])
        b = a + int(np.searchsorted(plan.cell[a:hi], c, side='right'))
        if max_trials:
            b = min(b, a + int(max_trials))
        out.append((a, b, c))
        a = b
    return out


@dataclass
class SweepResult:
    """Per-trial rows of a shard of the sweep on the HOST (pinned memory): what main.py keeps of a trial once its CSV rows are written."""
    plan: TrialPlan
    lo: int
    hi: int
    stats: np.ndarray        # (T_local, 3)  ||ISE||_2, ||IAE||_2, ||ITAE||_2
    status: np.ndarray       # (T_local,) int32
    k_done: np.ndarray       # (T_local,) int32
    pieces: list
    seconds: float = 0.0     # wall clock of the whole sweep, synchronised on both sides

    def rows(self):
        """(T_local, 5) fp64 [ISE, IAE, ITAE, status, k_done]."""
        return np.concatenate([self.stats, self.status[:, None].astype(float), self.k_done[:, None].astype(float)], axis=1)

    def cell_summary(self):
        from . import stats as _stats
        return _stats.cell_summary(self.stats, self.status, self.plan.cell[self.lo:self.hi])

    def gather(self, group=None, device=None):
        """The rows of ALL ranks in global trial order, (len(plan), 5), on every rank: the sweep's one collective, an all-gather
        of 40 B per trial.  ``device``: where the exchanged tensors live -- a cuda device for the nccl (RCCL) backend, None (host) for gloo."""
        import torch
        rows = torch.from_numpy(np.ascontiguousarray(self.rows()))
        if device is not None:
            rows = rows.to(device)
        return dist.gather_trial_rows(rows, len(self.plan), group).cpu().numpy()


def run_sweep(cfg, plant=None, cells=None, epoch=None, rank=0, world=1, want=(), lanes=0, device='cuda', max_trials=None,
              share_noise=True, on_piece=None, strict_pinv=False, latency=False, plan=None):
    """See _run_sweep.  The engine launches on the CURRENT device's current stream: the sweep runs with ``device`` made current, so that the events
    that order the row copies against the kernels are recorded on the stream the kernels are launched on."""
    import torch
    dev = torch.device(device)
    with torch.cuda.device(dev if dev.index is not None else torch.cuda.current_device()):
        return _run_sweep(cfg, plant, cells, epoch, rank, world, want, lanes, torch.device('cuda', torch.cuda.current_device()), max_trials, share_noise,
                          on_piece, strict_pinv, latency, plan)


def _run_sweep(cfg, plant, cells, epoch, rank, world, want, lanes, device, max_trials, share_noise, on_piece, strict_pinv, latency, plan):
    """The whole sweep of main.py:104-148 on this rank's GPU, cell after cell through ONE set of device buffers: for each piece (a cell, or
    ``max_trials`` trials of it) device seeding + noise generation (the T + 70 distinct streams where they alias), the closed-loop launch,
    and the per-trial [ISE, IAE, ITAE], status, k_done copied to pinned host memory on a second stream while the next piece computes.
    Device memory is that of the largest piece, whatever the sweep's size; results are bit-identical to ``run_batch`` over the same trials
    (same global seeds and jitter draws; a rank of ``world`` owns the contiguous global trials of dist.shard_range).

    ``want``: per-step streams to log on the device; they live until the next piece overwrites them, so ``on_piece(a, b, cell, out)`` --
    called after the piece has finished, with engine.closed_loop's dict -- is where a sink consumes them (it serialises the pipeline).
    ``plan``: a TrialPlan of this config made earlier (plan_trials takes 0.3 s for 786 432 trials; the GPU idles and clocks down meanwhile)."""
    import time
    import torch
    cfg = load_config(cfg)
    ex, est, nz = cfg['experiments'], cfg['estimator'], cfg['noise']
    method = Method[est['method']]
    if method == Method.ANALYTICAL:
        raise NotImplementedError('ANALYTICAL is not an estimator (and crashes in the reference: R is unbound, experiment.py:121)')
    plant = SyntheticPlant.ur10(ex['desired_f']) if plant is None else plant
    if plan is None:
        plan = plan_trials(cfg, cells, epoch)
    else:                                                          # a plan made earlier must be THIS sweep's: a stale one would silently run other seeds / cells
        n_cells = len(sweep_cells(NoiseType[nz['type']])) if cells is None else len(cells)
        per_cell = int(ex['epoch'] if epoch is None else epoch)
        first_seed = nz['seed']
        if len(plan) != n_cells * per_cell or len(plan.cells) != n_cells or (cells is not None and not np.array_equal(plan.cells, np.asarray(cells, float))) or \
                (first_seed is not None and len(plan) and int(plan.seed[0]) != int(first_seed)) or \
                (len(plan) and not ex['change_q_start'] and not np.array_equal(plan.q_start[0], np.asarray(ex['q_start'], float))):
            raise ValueError('run_sweep(plan=...): the plan does not belong to this config / cells / epoch')
    lo, hi = dist.shard_range(len(plan), rank, world)
    p = est['estimator_params']
    m, n = len(ex['desired_f']), plant.n_joints
    fp = engine.make_params(m, n, method.name, p.get('kernel_bw', 1.0), p.get('annealing', False), ex['dt'], ex['t_max'],
                            ex['ibvs_gain'], ex['desired_f'], p['initial_guess'], lanes, None, p.get('fpi_threshold', 0.1),
                            p.get('fpi_epoch_max', 1000))
    fp.reserved = (1 if strict_pinv else 0) | (2 if latency else 0)
    K = len(engine.loop_clock(ex['dt'], ex['t_max']))
    dev = torch.device(device)
    pieces = sweep_pieces(plan, lo, hi, max_trials)
    Tl = hi - lo
    host = dict(stats=torch.empty((Tl, 3), dtype=torch.float64).pin_memory(), status=torch.empty(Tl, dtype=torch.int32).pin_memory(),
                k_done=torch.empty(Tl, dtype=torch.int32).pin_memory())
    if not pieces:
        return SweepResult(plan, lo, hi, host['stats'].numpy(), host['status'].numpy(), host['k_done'].numpy(), pieces)
    Tmax = max(b - a for a, b, _ in pieces)
    noise_type = NoiseType[nz['type']]
    hold_cnt = int(nz['hold_time'] / ex['dt'])                      # main.py:137
    key = 'alpha' if noise_type == NoiseType.ALPHA_STABLE else 'rho'
    plant_struct = plant.to_struct()
    q0 = torch.as_tensor(plan.q_start[lo:hi].copy(), device=dev)
    x0 = None
    if not p['initial_guess']:
        x0 = torch.as_tensor(np.asarray(p['x0'], float).reshape(1, m * n).repeat(Tmax, 0), device=dev)
    seeds_dev = torch.as_tensor(np.ascontiguousarray(plan.seed[lo:hi], dtype=np.uint64).view(np.int64), device=dev)
    # two sets of per-trial outputs (a set is copied out while the next piece writes the other), one set of streams, one noise buffer
    S = Tmax + noise_device.SEED_STEP * (m - 1)
    noise_buf = torch.empty(K * max(S, m * Tmax), dtype=torch.float64, device=dev)
    streams = {k_: engine.alloc_stream(Tmax, K, c, 'kct', dev) for k_, c in (('x', m * n), ('err', m), ('q', n), ('f', m), ('dq', n)) if k_ in want}
    sets = []
    for _ in range(2):                                             # the streams are allocated ONCE and shared by both sets (7.5 GB of X per 65 536 trials)
        d = dict(streams)
        d.update(stats=torch.zeros((Tmax, 3), dtype=torch.float64, device=dev), status=torch.zeros(Tmax, dtype=torch.int32, device=dev),
                 k_done=torch.zeros(Tmax, dtype=torch.int32, device=dev))
        sets.append(d)
    main = torch.cuda.current_stream(dev)
    copier = _COPY_STREAMS.get(str(dev))                           # one per device, kept: the first use of a new stream costs ~4 ms on the host
    if copier is None:                                             # (queue creation), during which the GPU runs dry
        copier = _COPY_STREAMS[str(dev)] = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(copier):
            torch.zeros(1, device=dev)
    copied = []
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i, (a, b, c) in enumerate(pieces):
        T = b - a
        params = dict(nz['noise_params'])
        params[key] = float(plan.cells[c])                          # main.py:123-126
        if share_noise and noise_device.shares_streams(noise_type, nz['hold'], plan.seed[a:b]):
            St = T + noise_device.SEED_STEP * (m - 1)
            noise = noise_device.generate_shared(noise_type, params, int(plan.seed[a]), T, m, K, out=noise_buf[:K * St].view(K, St), device=dev)[1]
        else:
            noise = noise_buf[:K * m * T].view(K, m, T)
            noise_device.generate(noise_type, params, seeds_dev[a - lo:b - lo], m, K, nz['hold'], hold_cnt, 'kct', out=noise, device=dev)
        if i >= 2:
            main.wait_event(copied[i - 2])                          # this set's previous rows have left the device
        out = engine.closed_loop(fp, plant_struct, q0[a - lo:b - lo], noise, None if x0 is None else x0[:T], want=want, reuse=sets[i % 2])
        done = torch.cuda.Event()
        done.record(main)
        copier.wait_event(done)
        with torch.cuda.stream(copier):
            for k_ in ('stats', 'status', 'k_done'):
                host[k_][a - lo:b - lo].copy_(out[k_], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copier)
        copied.append(ev)
        if on_piece is not None:
            done.synchronize()
            on_piece(a, b, c, out)
    torch.cuda.synchronize(dev)
    seconds = time.perf_counter() - t0
    return SweepResult(plan, lo, hi, host['stats'].numpy(), host['status'].numpy(), host['k_done'].numpy(), pieces, seconds)


CSV_COLUMNS = (['experiment_id', 'status', 'rho', 't'] + [f'q_{i}' for i in range(1, 7)] +
               ['camera_x', 'camera_y', 'camera_z', 'camera_roll', 'camera_pitch', 'camera_yaw'] +
               [f'f_{i}' for i in range(1, 9)] + [f'desired_f_{i}' for i in range(1, 9)] + [f'noise_{i}' for i in range(1, 9)] + ['kernel_bw'])


def write_results_csv(result, cfg, plant, path):
    """results.csv in the reference's long format (main.py:152-196): one row per logged step, 41 columns.  Needs the
    'q' and 'f' streams; meant for sweeps of the reference's size (~10^3 trials) -- larger runs keep tensors/npz."""
    import pandas as pd
    q = result.streams['q'].cpu().numpy()
    f = result.streams['f'].cpu().numpy()
    noise = result.noise.cpu().numpy()
    status, k_done = result.status.cpu().numpy(), result.k_done.cpu().numpy()
    desired = np.asarray(cfg['experiments']['desired_f'], float)
    method, p = Method[cfg['estimator']['method']], cfg['estimator']['estimator_params']
    header = not os.path.exists(path)
    for j in range(result.hi - result.lo):
        k = int(k_done[j])
        cam = plant.camera_pose_batch(q[:k, :, j]) if k else np.zeros((0, 6))   # computePose (ur10_simulation.py:151-163)
        cols = {'experiment_id': result.lo + j, 'status': ExperimentStatus(int(status[j])), 'rho': result.plan.value[result.lo + j],
                't': result.t[:k]}
        cols.update({f'q_{i + 1}': q[:k, i, j] for i in range(6)})
        cols.update({name: cam[:, i] for i, name in enumerate(CSV_COLUMNS[10:16])})
        cols.update({f'f_{i + 1}': f[:k, i, j] for i in range(8)})
        cols.update({f'desired_f_{i + 1}': np.full(k, desired[i]) for i in range(8)})
        cols.update({f'noise_{i + 1}': noise[:k, i, j] for i in range(8)})
        cols['kernel_bw'] = bandwidth_log(method, k, p.get('kernel_bw', 1.0), p.get('annealing', False), cfg['experiments']['dt'],
                                          cfg['experiments']['t_max'])                      # -1 unless MCKF (experiment.py:330)
        pd.DataFrame(data=cols).to_csv(path, mode='a', index=False, header=header)
        header = False


def write_results_parquet(result, cfg, plant, path, trials_per_group=2048):
    """The reference's results table (main.py:152-196: one row per logged step, the 41 columns of results.csv, same names, same order) as a
    Parquet file, for sweeps where the long-format CSV is impractical (65 536 trials x 299 rows x 41 columns is 6.7 GB of text).  Streams:
    ``trials_per_group`` trials at a time are copied from the device, expanded and written as one row group, so the host never holds more
    than one group (2 048 trials = 612 352 rows = 200 MB).  ``status`` is the reference's string (``ExperimentStatus.SUCCESS``), dictionary
    encoded; rows at and after a FAILed trial's k_done are dropped as the reference trims them (experiment.py:345-352).  Needs the 'q' and
    'f' streams.  Returns the number of rows written."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    desired = np.asarray(cfg['experiments']['desired_f'], float)
    method, p = Method[cfg['estimator']['method']], cfg['estimator']['estimator_params']
    status_all, k_all = result.status.cpu().numpy(), result.k_done.cpu().numpy()
    T, K = result.hi - result.lo, len(result.t)
    bw_full = bandwidth_log(method, K, p.get('kernel_bw', 1.0), p.get('annealing', False), cfg['experiments']['dt'], cfg['experiments']['t_max'])
    names = {0: str(ExperimentStatus(0)), 1: str(ExperimentStatus(1))}
    writer, rows_written = None, 0
    try:
        for a in range(0, T, trials_per_group):
            b = min(T, a + trials_per_group)
            q = result.streams['q'][:, :, a:b].cpu().numpy()               # [K][6][chunk]
            f = result.streams['f'][:, :, a:b].cpu().numpy()
            nz = result.noise[:, :, a:b].cpu().numpy()
            keep = (np.arange(K)[None, :] < k_all[a:b, None])              # (chunk, K): logged rows
            tt, kk = np.nonzero(keep)                                      # trial-major, step-minor: the reference's row order
            cam = plant.camera_pose_batch(np.moveaxis(q, 1, 2)[kk, tt])     # (rows, 6)
            cols = {'experiment_id': (result.lo + a + tt).astype(np.int64),
                    'status': pa.array([names[int(v)] for v in status_all[a:b]]).dictionary_encode().take(pa.array(tt)),
                    'rho': result.plan.value[result.lo + a + tt], 't': np.asarray(result.t)[kk]}
            cols.update({f'q_{i + 1}': q[kk, i, tt] for i in range(6)})
            cols.update({name: cam[:, i] for i, name in enumerate(CSV_COLUMNS[10:16])})
            cols.update({f'f_{i + 1}': f[kk, i, tt] for i in range(8)})
            cols.update({f'desired_f_{i + 1}': np.full(len(tt), desired[i]) for i in range(8)})
            cols.update({f'noise_{i + 1}': nz[kk, i, tt] for i in range(8)})
            cols['kernel_bw'] = bw_full[kk]
            table = pa.table({name: cols[name] for name in CSV_COLUMNS})
            if writer is None:
                writer = pq.ParquetWriter(path, table.schema, compression='zstd')
            writer.write_table(table)
            rows_written += len(tt)
    finally:
        if writer is not None:
            writer.close()
    return rows_written


def save_results_npz(result, cfg, path, streams=True):
    """Compact sink for sweeps too large for the long-format CSV (65 536 trials x 299 rows x 41 columns is 6.7 GB of text):
    per-trial rows (experiment_id, swept value, seed, status, k_done, ISE / IAE / ITAE norms, q_start), the clock, the config,
    and -- if ``streams`` -- whatever per-step tensors the run kept, trial-fastest as on the device ([step][component][trial]).
    For a FAILed trial only the first ``k_done`` rows of a stream are meaningful (the reference trims its logs there, experiment.py:345-352)."""
    data = {'experiment_id': np.arange(result.lo, result.hi), 'cell': result.plan.cell[result.lo:result.hi],
            'rho': result.plan.value[result.lo:result.hi], 'seed': result.plan.seed[result.lo:result.hi],
            'q_start': result.plan.q_start[result.lo:result.hi], 't': np.asarray(result.t),
            'status': result.status.cpu().numpy(), 'k_done': result.k_done.cpu().numpy(), 'stats': result.stats.cpu().numpy(),
            'stats_columns': np.array(['ise', 'iae', 'itae']), 'config': json.dumps(cfg)}
    if streams:
        for name, tensor in result.streams.items():
            data['stream_' + name] = tensor.cpu().numpy()
    np.savez_compressed(path, **data)
    return path

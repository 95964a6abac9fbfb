"""The multi-rank path ON THE HIP KERNEL (BASELINE config 4's code path; main.py:127-148 is the loop being sharded).

A 1-GPU box cannot hold eight ranks on eight cards, but it can run every line of the sharded path: fresh child processes (one per rank)
share the card, each runs its contiguous shard of the global trial enumeration through `batch.run_batch(cfg, rank=r, world=W)` and the
ranks all-gather the per-trial [ISE, IAE, ITAE, status] rows.  Because trials are independent and seeds / jitter draws follow the
GLOBAL trial index, the gathered table must equal a one-rank run of the same plan bit for bit.  A second test launches bench.py under
torch.distributed.run with --force-dist so that RCCL initialisation, the barrier and the all-gather of cuda tensors inside the timed
region run once under the driver.  No scaling number is claimed from either."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = os.path.join(ROOT, 'tests', '_dist_gpu_worker.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def _run_ranks(world, total, backend, out_dir, sweep=False):
    for attempt in range(3):                                                  # a port found free can be taken before the ranks bind it: try another
        port = _free_port()
        env = dict(_env(), UVS_TEST_SWEEP='1') if sweep else _env()
        procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(total), backend, str(out_dir)], env=env, cwd=ROOT,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        logs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            logs.append(out)
        if any(p.returncode != 0 for p in procs) and any('ddress already in use' in log or 'EADDRINUSE' in log for log in logs) and attempt < 2:
            continue
        for r, p in enumerate(procs):
            assert p.returncode == 0, f'rank {r} failed:\n{logs[r][-3000:]}'
        return [np.load(os.path.join(out_dir, f'rank{r}.npz')) for r in range(world)]


def _torchrun_bench(nproc, *bench_args):
    """bench.py under torch.distributed.run with `nproc` ranks (sharing the box's one GPU when nproc > 1); returns the parsed JSON line."""
    for attempt in range(3):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
               os.path.join(ROOT, 'bench.py'), '--gpus', str(nproc), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-side', '--no-replay'] + list(bench_args)
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=ROOT, env=_env())
        if res.returncode != 0 and ('ddress already in use' in res.stdout + res.stderr) and attempt < 2:
            continue
        assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == 1
        return json.loads(lines[0])


def _single_rank(total):
    import uvs_amd
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _dist_gpu_worker import sweep_config
    res = uvs_amd.batch.run_batch(sweep_config(total), cells=[1.5], want=())
    return uvs_amd.dist.pack_rows(res.stats, res.status).cpu().numpy(), res.k_done.cpu().numpy()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world,total', [(2, 1001), (3, 100)])
def test_sharded_sweep_on_the_hip_kernel_equals_one_rank(tmp_path, world, total):
    """gloo ranks sharing the one GPU, ragged shards (odd totals): gathered rows bit-identical to the 1-rank sweep on every rank."""
    import uvs_amd
    parts = _run_ranks(world, total, 'gloo', tmp_path)
    single, k_single = _single_rank(total)
    assert single.shape == (total, 4) and np.all(np.isfinite(single)) and np.all(single[:, :3] > 0)
    for r, z in enumerate(parts):
        lo, hi = uvs_amd.dist.shard_range(total, r, world)
        assert (int(z['lo']), int(z['hi'])) == (lo, hi)
        assert np.array_equal(z['rows'], single), f'rank {r}: gathered rows differ from the one-rank sweep'
        assert np.array_equal(z['k_done'], k_single[lo:hi])
    assert sum(int(z['hi']) - int(z['lo']) for z in parts) == total


@pytest.mark.timeout(900)
def test_sharded_run_sweep_and_gather_equal_one_rank(tmp_path):
    """batch.run_sweep on 3 gloo ranks sharing the GPU (ragged shards, pieces of 64 trials) + SweepResult.gather: every rank ends up with
    the one-rank table, k_done included."""
    import uvs_amd
    total = 301
    parts = _run_ranks(3, total, 'gloo', tmp_path, sweep=True)
    single, k_single = _single_rank(total)
    for r, z in enumerate(parts):
        lo, hi = uvs_amd.dist.shard_range(total, r, 3)
        assert (int(z['lo']), int(z['hi'])) == (lo, hi)
        assert np.array_equal(z['rows'], single) and np.array_equal(z['k_done_all'], k_single) and np.array_equal(z['k_done'], k_single[lo:hi])


@pytest.mark.timeout(900)
def test_one_rank_rccl_gather_equals_plain_run(tmp_path):
    """The same worker on the nccl (= RCCL) backend, world size 1: process-group init on the GPU and an all_gather of cuda tensors."""
    total = 257
    (z,) = _run_ranks(1, total, 'nccl', tmp_path)
    single, _ = _single_rank(total)
    assert np.array_equal(z['rows'], single)


@pytest.mark.timeout(900)
def test_bench_force_dist_under_torchrun():
    """bench.py's multi-rank branch (process group, barrier, gather inside the timed region, max-over-ranks) on RCCL with one rank."""
    d = _torchrun_bench(1, '--force-dist', '--trials', '4096')
    assert d['n_gpus'] == 1 and d['scaling'] == 'weak' and d['config']['trials_total'] == 4096 and d['config']['failed_trials'] == 0
    assert d['config']['ranks_seen'] == 1 and d['multi_gpu']['gather_inside_timed_region'] is True and d['multi_gpu']['backend'] == 'nccl'
    assert d['multi_gpu']['gather_ms'] >= 0 and d['multi_gpu']['kernel_ms_avg_over_ranks']['max'] >= d['multi_gpu']['kernel_ms_avg_over_ranks']['min'] > 0
    assert d['value'] > 0 and abs(d['value'] - 4096 * 299 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6


@pytest.mark.timeout(900)
@pytest.mark.parametrize('scaling', ['strong', 'weak'])
def test_bench_two_ranks_by_name(scaling):
    """bench.py --gpus 2 on two gloo ranks sharing the GPU: the strong series keeps the total (ragged shards of an odd total), the weak
    series gives every rank the size; trial totals, ranks seen and the whole-job throughput formula."""
    d = _torchrun_bench(2, '--backend', 'gloo', '--trials', '3001', '--scaling', scaling)
    total = 3001 if scaling == 'strong' else 6002
    assert d['n_gpus'] == 2 and d['scaling'] == scaling and d['config']['trials_total'] == total and d['config']['ranks_seen'] == 2
    assert d['config']['trials_rank0'] == (1501 if scaling == 'strong' else 3001) and d['config']['failed_trials'] == 0
    assert abs(d['value'] - total * 299 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    assert d['multi_gpu']['backend'] == 'gloo' and d['multi_gpu']['gather_ms'] > 0


@pytest.mark.timeout(900)
def test_bench_three_ranks_strong_series():
    """Three gloo ranks sharing the GPU (the box allows six processes on the card: three ranks + the launcher + this test process leave a
    margin): north_star's strong series at a total the card holds three times over, ragged shards (4 099 = 1 367 + 1 366 + 1 366)."""
    d = _torchrun_bench(3, '--backend', 'gloo', '--trials', '4099', '--scaling', 'strong')
    assert d['n_gpus'] == 3 and d['scaling'] == 'strong' and d['config']['trials_total'] == 4099 and d['config']['ranks_seen'] == 3
    assert d['config']['trials_rank0'] == 1367 and d['config']['failed_trials'] == 0
    assert abs(d['value'] - 4099 * 299 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    assert d['multi_gpu']['kernel_ms_avg_over_ranks']['max'] >= d['multi_gpu']['kernel_ms_avg_over_ranks']['min'] > 0


def _plain_bench(*bench_args, expect_rc0=True):
    """`python bench.py ...` exactly as the driver types it (no torchrun, no WORLD_SIZE): for --gpus N > 1 the process starts its own ranks."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + list(bench_args)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=ROOT, env=_env())
    if not expect_rc0:
        return res
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and res.stdout.strip() == lines[0]                 # one JSON line and nothing else on stdout
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_plain_bench_gpus_2_launches_its_own_ranks():
    """the driver's command shape with --gpus 2 on a 1-GPU box.  The parent starts two fresh ranks before touching the GPU;
    with one visible card the ranks share it and the gather falls back to gloo (said so in the line).  Default series: weak (per-GPU work
    fixed); north_star's strong series of the same job rides along as multi_gpu.strong_series."""
    d = _plain_bench('--gpus', '2', '--steps', '2', '--warmup', '1', '--no-side', '--no-cpu-baseline', '--trials', '4096')
    assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2 and d['scaling'] == 'weak' and d['config']['trials_total'] == 8192
    assert d['config']['trials_rank0'] == 4096 and d['config']['failed_trials'] == 0
    assert d['multi_gpu']['gather_ms'] > 0 and d['multi_gpu']['gather_inside_timed_region'] is True
    assert d['multi_gpu']['backend'] == 'gloo' and 'share' in d['multi_gpu']['backend_note']
    assert abs(d['value'] - 8192 * 299 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    s = d['multi_gpu']['strong_series']
    assert s['scaling'] == 'strong' and s['trials_total'] == 4096 and s['trials_rank0'] == 2048 and s['n_gpus'] == 2 and s['failed_trials'] == 0
    assert abs(s['value'] - 4096 * 299 / (s['ms_per_step'] * 1e-3)) / s['value'] < 1e-6


@pytest.mark.timeout(900)
def test_plain_bench_config_4_gpus_2():
    d = _plain_bench('--config', '4', '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-side', '--no-cpu-baseline', '--trials', '8192')
    assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2 and d['scaling'] == 'strong' and d['config']['trials_total'] == 8192
    assert d['config']['trials_rank0'] == 4096 and 'config 4' in d['config']['workload'] and d['multi_gpu']['gather_ms'] > 0
    assert d['multi_gpu']['strong_series'] is None


@pytest.mark.timeout(900)
def test_plain_bench_relays_a_failing_rank():
    """A rank that dies (here: every rank, on an impossible request) makes the self-launching parent exit non-zero without a JSON line."""
    res = _plain_bench('--gpus', '2', '--steps', '1', '--warmup', '0', '--no-side', '--no-cpu-baseline', '--scaling', 'strong', '--trials', '1', expect_rc0=False)   # fewer trials than ranks
    assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert 'fewer trials than ranks' in res.stderr


@pytest.mark.timeout(900)
def test_bench_config_4_by_name():
    """--config 4 is a total (BASELINE config 4: 1 048 576 trials over all ranks); here its code path on two gloo ranks with the total
    overridden to something the shared card holds, and the refusal of a weak form."""
    d = _torchrun_bench(2, '--backend', 'gloo', '--config', '4', '--trials', '8192')
    assert d['scaling'] == 'strong' and d['config']['trials_total'] == 8192 and d['config']['trials_rank0'] == 4096 and 'config 4' in d['config']['workload']
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', '4', '--scaling', 'weak'], capture_output=True, text=True, cwd=ROOT, env=_env())
    assert res.returncode != 0 and 'no weak form' in res.stderr


@pytest.mark.timeout(900)
def test_config_4_shard_of_rank_3_of_8():
    """BASELINE config 4 at its size, one rank's share: rank 3 of 8 owns the global trials 393 216 ... 524 287 of the 1 048 576
    (main.py:121-148 enumerates them; seeds 123456 + t, the t-th jitter draws).  The shard runs on the one GPU exactly as that rank would
    run it; sampled trials are checked against the C oracle on host-generated noise of the same GLOBAL indices, and the same index range
    run as ranks 6 and 7 of a 16-way split returns bit-identical rows (partition invariance)."""
    import torch
    import uvs_amd
    from oracle import c_oracle
    import bench
    total, K = 1048576, 299
    cfg = bench.config2()
    cfg['experiments']['epoch'] = total
    res = uvs_amd.batch.run_batch(cfg, cells=[1.5], rank=3, world=8, want=('err',))
    assert (res.lo, res.hi) == (393216, 524288) and res.stats.shape == (131072, 3)
    rows = uvs_amd.dist.pack_rows(res.stats, res.status).cpu().numpy()
    k_done = res.k_done.cpu().numpy()
    plan = res.plan
    assert plan.seed[res.lo] == 123456 + 393216 and len(plan) == total
    sample = np.array([393216, 393217, 400000, 458752, 524287])
    noise = np.zeros((len(sample), K, 8))
    for i, t in enumerate(sample):
        uvs_amd.batch.trial_noise(cfg, plan, int(t), int(t) + 1, K, noise[i:i + 1])
    ref = c_oracle.closed_loop_batch(plan.q_start[sample], noise, cfg['experiments']['desired_f'])
    err = res.streams['err']
    for i, t in enumerate(sample):
        j = int(t) - res.lo
        assert int(ref['status'][i]) == int(rows[j, 3]) == 0 and int(ref['k_done'][i]) == int(k_done[j]) == K
        a = err[:, :, j].cpu().numpy()
        assert np.abs(a - ref['err'][i]).max() / np.abs(ref['err'][i]).max() <= 1e-8, int(t)
        assert np.abs(rows[j, :3] - ref['stats'][i]).max() / ref['stats'][i].max() <= 1e-8
    del err, res
    torch.cuda.empty_cache()
    halves = [uvs_amd.batch.run_batch(cfg, cells=[1.5], rank=r, world=16, want=()) for r in (6, 7)]
    assert (halves[0].lo, halves[1].hi) == (393216, 524288)
    again = np.concatenate([uvs_amd.dist.pack_rows(h.stats, h.status).cpu().numpy() for h in halves])
    assert np.array_equal(again, rows) and np.array_equal(np.concatenate([h.k_done.cpu().numpy() for h in halves]), k_done)

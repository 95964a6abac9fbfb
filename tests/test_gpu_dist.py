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


def _run_ranks(world, total, backend, out_dir):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(total), backend, str(out_dir)], env=_env(), cwd=ROOT,
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
    for r, p in enumerate(procs):
        assert p.returncode == 0, f'rank {r} failed:\n{logs[r][-3000:]}'
    return [np.load(os.path.join(out_dir, f'rank{r}.npz')) for r in range(world)]


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
def test_one_rank_rccl_gather_equals_plain_run(tmp_path):
    """The same worker on the nccl (= RCCL) backend, world size 1: process-group init on the GPU and an all_gather of cuda tensors."""
    total = 257
    (z,) = _run_ranks(1, total, 'nccl', tmp_path)
    single, _ = _single_rank(total)
    assert np.array_equal(z['rows'], single)


@pytest.mark.timeout(900)
def test_bench_force_dist_under_torchrun():
    """bench.py's multi-rank branch (bench.py: process group, barrier, gather inside the timed region, max-over-ranks) on RCCL with one rank."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '2', '--warmup', '1', '--trials', '4096', '--no-cpu-baseline']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=ROOT, env=_env())
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['scaling'] == 'weak' and d['config']['trials_per_gpu'] == 4096 and d['config']['failed_trials'] == 0
    assert d['value'] > 0 and abs(d['value'] - 4096 * 299 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6

"""The multi-GPU path on CPU: two gloo ranks shard a sweep exactly like two GPUs would (contiguous global trial ranges, same
global seeds / jitter draws) and all-gather the per-trial [ISE, IAE, ITAE, status] rows.  The estimator stand-in is the oracle --
the HIP kernel cannot run here; what is under test is sharding, noise/seed assignment and the gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _shard_rows(cfg, plan, lo, hi, K):
    """Per-trial rows of this shard computed by the C oracle on the product's noise / jitter assignment."""
    import uvs_amd
    from oracle import c_oracle
    noise = np.zeros((hi - lo, K, 8))
    uvs_amd.batch.trial_noise(cfg, plan, lo, hi, K, noise)
    out = c_oracle.closed_loop_batch(plan.q_start[lo:hi], noise, cfg['experiments']['desired_f'], 'GMCKF', 10.0, False, 0.05, 15, 0.2)
    return np.concatenate([out['stats'], out['status'][:, None].astype(float)], axis=1)


def _config(epoch):
    import json
    cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_reference.json')))
    cfg['estimator']['method'] = 'GMCKF'
    cfg['experiments']['epoch'] = epoch
    return cfg


def _worker(rank, world, port, total, K, ret):
    import sys
    sys.path.insert(0, ROOT)
    import uvs_amd
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    td.init_process_group('gloo', rank=rank, world_size=world)
    cfg = _config(total)
    plan = uvs_amd.batch.plan_trials(cfg, cells=[1.5])
    lo, hi = uvs_amd.dist.shard_range(len(plan), rank, world)
    rows = torch.as_tensor(_shard_rows(cfg, plan, lo, hi, K))
    full = uvs_amd.dist.gather_trial_rows(rows, len(plan))
    ret[rank] = (lo, hi, full.numpy())
    td.destroy_process_group()


def test_shard_ranges_cover_and_balance():
    import uvs_amd
    for total, world in ((10, 3), (65536, 8), (7, 8), (1048576, 8)):
        spans = [uvs_amd.dist.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sweep_equals_single_rank():
    import uvs_amd
    total, K, world = 11, 60, 2                                   # odd total: ragged shards (6 + 5)
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), total, K, ret), nprocs=world, join=True)
    cfg = _config(total)
    plan = uvs_amd.batch.plan_trials(cfg, cells=[1.5])
    single = _shard_rows(cfg, plan, 0, total, K)
    assert ret[0][:2] == (0, 6) and ret[1][:2] == (6, 11)
    for r in range(world):
        assert np.array_equal(ret[r][2], single)                  # bit-for-bit: trials are independent of the partition
    pack = uvs_amd.dist.pack_rows(torch.as_tensor(single[:, :3]), torch.as_tensor(single[:, 3].astype(np.int32)))
    assert torch.equal(pack, torch.as_tensor(single))

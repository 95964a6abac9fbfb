"""Child process of tests/test_gpu_dist.py: one rank of a sharded Monte-Carlo sweep on the HIP kernel.

    python tests/_dist_gpu_worker.py RANK WORLD PORT TOTAL BACKEND OUT_DIR

Every rank runs `batch.run_batch(cfg, rank=r, world=W)` (UVS_TEST_SWEEP=1: `batch.run_sweep`) on the (shared) GPU, packs its per-trial [ISE, IAE, ITAE, status] rows and
all-gathers them (`dist.gather_trial_rows`), exactly what bench.py / a real N-GPU sweep does; the gathered table and the shard bounds
are written to OUT_DIR/rank<r>.npz for the parent test to compare with a one-rank run.  Started as a fresh interpreter (never a fork
or re-exec of a process that already touched the GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sweep_config(total):
    cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_reference.json')))
    cfg['estimator']['method'] = 'GMCKF'
    cfg['experiments']['epoch'] = total
    return cfg


def main():
    rank, world, port, total = (int(a) for a in sys.argv[1:5])
    backend, out_dir = sys.argv[5], sys.argv[6]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    import torch.distributed as td
    import uvs_amd
    torch.cuda.set_device(0)                                      # every rank on the box's one GPU
    if backend == 'nccl':
        td.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    else:
        td.init_process_group('gloo', rank=rank, world_size=world)
    cfg = sweep_config(total)
    if os.environ.get('UVS_TEST_SWEEP'):                          # the same shard through batch.run_sweep (pieces of 64 trials) and SweepResult.gather
        sw = uvs_amd.batch.run_sweep(cfg, cells=[1.5], rank=rank, world=world, max_trials=64)
        full = sw.gather(device=torch.device('cuda', 0) if backend == 'nccl' else None)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), lo=sw.lo, hi=sw.hi, rows=full[:, :4], k_done=sw.k_done, k_done_all=full[:, 4])
        td.barrier()
        td.destroy_process_group()
        return
    res = uvs_amd.batch.run_batch(cfg, cells=[1.5], rank=rank, world=world, want=())
    rows = uvs_amd.dist.pack_rows(res.stats, res.status)
    full = uvs_amd.dist.gather_trial_rows(rows if backend == 'nccl' else rows.cpu(), len(res.plan))
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), lo=res.lo, hi=res.hi, rows=full.cpu().numpy(), k_done=res.k_done.cpu().numpy())
    td.barrier()
    td.destroy_process_group()


if __name__ == '__main__':
    main()

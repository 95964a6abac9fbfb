"""Multi-GPU sharding of a Monte-Carlo batch: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Trials are independent (main.py:127-148: own X, P, noise streams, q_start), so the data path has no collective:
rank r owns the contiguous global trials [lo, hi) and runs them exactly as a single GPU would (same global seeds, same
jitter draws), which makes a G-GPU sweep bit-identical to the 1-GPU sweep.  The only exchange is one end-of-run
all-gather of the per-trial [ISE, IAE, ITAE, status] rows (32 B per trial) over xGMI.
"""


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) of global trial indices for ``rank``."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_trial_rows(local_rows, total, group=None):
    """All-gather per-trial rows.  ``local_rows``: (hi - lo, C) tensor on this rank's device (cuda for nccl, cpu for gloo).
    Returns the (total, C) tensor in global trial order on every rank."""
    import torch
    import torch.distributed as td
    world, rank = td.get_world_size(group), td.get_rank(group)
    sizes = [shard_range(total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    padded = torch.zeros((width, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    padded[:local_rows.shape[0]] = local_rows
    parts = [torch.empty_like(padded) for _ in range(world)]
    td.all_gather(parts, padded, group=group)
    return torch.cat([parts[r][:hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def pack_rows(stats, status):
    """(T, 3) stats + (T,) int status -> (T, 4) fp64 rows [ISE, IAE, ITAE, status]."""
    import torch
    return torch.cat([stats, status.to(stats.dtype).unsqueeze(1)], dim=1)

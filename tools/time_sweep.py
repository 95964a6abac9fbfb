#!/usr/bin/env python3
"""batch.run_sweep on the reference's 12-cell sweep at 65 536 trials per cell: wall time per cell, stats only and with the per-step streams.
usage (GPU box): python tools/time_sweep.py [trials per cell] [method] [variant index 0-5: only that one]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import uvs_amd  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = bench.config2()
cfg['experiments']['epoch'] = T
if len(sys.argv) > 2:
    cfg['estimator']['method'] = sys.argv[2]
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
plan = uvs_amd.batch.plan_trials(cfg)
i = -1
for want in ((), ('err', 'q', 'f'), ('x', 'err', 'q')):
    for share in (True, False):
        i += 1
        if only is not None and i != only:
            continue
        uvs_amd.batch.run_sweep(cfg, want=want, share_noise=share, plan=plan)            # warm-up (allocator, tables)
        for _ in range(3):
            r = uvs_amd.batch.run_sweep(cfg, want=want, share_noise=share, plan=plan)
            print(f'{cfg["estimator"]["method"]} want={want} shared_noise={share}: {r.seconds * 1e3 / 12:.3f} ms per cell, {int(r.k_done.sum()) / r.seconds / 1e9:.2f} G updates/s, '
                  f'failed {int((r.status != 0).sum())}', flush=True)

#!/usr/bin/env python3
"""The reference's Monte-Carlo experiment (main.py with config.json: 12 values of alpha x `epoch` trials, alpha-stable noise)
for the RMCKF (Method.GMCKF) on the synthetic UR10 plant, as one GPU sweep.  Prints ITAE mean / std / median per alpha, the
quantity plotted in results/results1.fig of the reference.

    python examples/sweep_alpha.py [--epoch 100] [--method GMCKF|KF|IMCCKF|MCKF] [--csv results.csv]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvs_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config.json'))
    ap.add_argument('--epoch', type=int, default=None)
    ap.add_argument('--method', default='GMCKF')
    ap.add_argument('--csv', default=None, help='also write the reference-format results.csv (small sweeps only)')
    args = ap.parse_args()
    cfg = json.load(open(args.config))
    cfg.pop('_about', None)
    cfg['estimator']['method'] = args.method
    if args.csv:                                                 # every trial's per-step streams stay on the device for the CSV writer: one grid
        want = ('err', 'q', 'f')
        uvs_amd.batch.run_batch(cfg, epoch=1, want=want)        # warm-up: loads the code object, uploads the ziggurat tables
        res = uvs_amd.batch.run_batch(cfg, epoch=args.epoch, want=want)
        summary = uvs_amd.stats.cell_summary(res.stats.cpu().numpy(), res.status.cpu().numpy(), res.plan.cell)
        print(f'{len(res.plan)} trials in {res.seconds * 1e3:.2f} ms of kernel time ({args.method})')
    else:                                                        # statistics only: cell after cell through one set of buffers, any epoch
        uvs_amd.batch.run_sweep(cfg, epoch=1)
        res = uvs_amd.batch.run_sweep(cfg, epoch=args.epoch)
        summary = res.cell_summary()
        print(f'{len(res.plan)} trials in {res.seconds * 1e3:.2f} ms end to end: seeding, noise, closed loop, rows on the host ({args.method})')
    for c, row in summary.items():
        print(f'alpha = {res.plan.cells[c]:.4f}  ok {row["success"]:4d}/{row["trials"]:4d}  ITAE mean {row["itae_mean"]:12.1f}  std {row["itae_std"]:12.1f}  median {row["itae_median"]:12.1f}')
    if args.csv:
        uvs_amd.batch.write_results_csv(res, cfg, uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']), args.csv)
        print('wrote', args.csv)


if __name__ == '__main__':
    main()

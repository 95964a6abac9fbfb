#!/usr/bin/env python3
"""profiles/rNN/scale_expectation.json: what `python bench.py --gpus N` should print at N = 1, 2, 4, 8 on one 8-GPU MI355X node, stated BEFORE anybody
has measured it (no multi-GPU node has reached this project in six rounds), so that the first SCALE_rNN.json is checked against a prediction.

Inputs: the one-GPU bench object of the round (UVS_BENCH_FULL_JSON: headline ms_per_step and multi_gpu.shard_model, the kernel time of the shards a rank of
the strong series owns) and a model of the one collective of the path, the all-gather of the per-trial [ISE, IAE, ITAE, status] rows (32 B per trial,
SURVEY 8e), which bench.py issues after every launch, inside the timed region and not overlapped with it:

    t_gather(N, trials_total) = t_launch + (N - 1) / N * 32 B * trials_total / busbw          (ring / direct all-gather over xGMI)

with t_launch = 30 us (RCCL kernel launch + synchronisation of a small collective) and busbw = 100 GB/s at 2-4 MB, 150 GB/s at 8-17 MB, 200 GB/s at
34 MB (7 links x ~76 GB/s per direction per GPU; messages this small run far below the link peak).  A pessimistic column doubles the gather and adds
the clock spread of the pool (the slowest of N cards sets the time: 2.05-2.15 GHz under the 1 400 W cap, i.e. up to + 2.5 % at N = 8).

usage: tools/scale_expectation.py gpurun_out/<call>/bench_full.json profiles/r06/scale_expectation.json"""
import json
import sys


def busbw(nbytes):
    return 100e9 if nbytes < 6e6 else (150e9 if nbytes < 25e6 else 200e9)


def gather_ms(n, trials_total):
    if n == 1:
        return 0.0
    nbytes = 32 * trials_total
    return (30e-6 + (n - 1) / n * nbytes / busbw(nbytes)) * 1e3


def main():
    src, dst = sys.argv[1], sys.argv[2]
    d = json.load(open(src))
    k1 = d['roofline']['avg_kernel_ms']
    step1 = d['ms_per_step']
    host = step1 - k1                                            # what a step costs beyond its kernel on one GPU (launch + loop overhead)
    sm = d['multi_gpu']['shard_model']['shards']
    T, K = 65536, 299
    spread = {1: 0.0, 2: 0.01, 4: 0.02, 8: 0.025}
    out = {'source': src, 'n1': {'ms_per_step': step1, 'kernel_ms': k1, 'value': d['value']},
           'model': __doc__.split('Inputs:')[1].split('usage:')[0].strip(), 'weak': {}, 'strong': {}, 'config4': {}}
    for n in (1, 2, 4, 8):
        g = gather_ms(n, n * T)
        ms = k1 + host + g
        worst = (k1 * (1 + spread[n])) + host + 2 * g
        out['weak'][f'N={n}'] = {'trials_total': n * T, 'gather_ms': round(g, 4), 'ms_per_step': round(ms, 3), 'value_updates_per_s': n * T * K / (ms * 1e-3),
                                 'efficiency_vs_n1': round(step1 / ms, 4), 'pessimistic_ms_per_step': round(worst, 3), 'pessimistic_efficiency': round(step1 / worst, 4)}
        ks = k1 if n == 1 else sm[f'N={n}']['default_mapping']['kernel_ms']
        g = gather_ms(n, T)
        ms = ks + host + g
        worst = ks * (1 + spread[n]) + host + 2 * g
        out['strong'][f'N={n}'] = {'trials_total': T, 'trials_per_rank': T // n, 'shard_kernel_ms': round(ks, 4), 'gather_ms': round(g, 4), 'ms_per_step': round(ms, 3),
                                   'speedup_vs_n1': round(step1 / ms, 3), 'pessimistic_ms_per_step': round(worst, 3), 'pessimistic_speedup': round(step1 / worst, 3)}
    g = gather_ms(8, 16 * T)
    ms = 2 * k1 + host + g
    out['config4'] = {'trials_total': 16 * T, 'trials_per_rank': 2 * T, 'n_gpus': 8, 'kernel_ms': round(2 * k1, 3), 'gather_ms': round(g, 4), 'ms_per_step': round(ms, 3),
                      'value_updates_per_s': 16 * T * K / (ms * 1e-3), 'note': '131 072 trials per rank = four rounds of wavefronts: twice the 65 536-trial kernel time (measured on one GPU: config-4 shard test)'}
    out['reading'] = ('weak series: per-GPU work fixed, the only added cost is the all-gather, so efficiency stays >= 0.95 (>= 0.90 pessimistic) to N = 8; strong series '
                      "(north_star's 65 536 trials in total): a trial is a 299-step serial chain and a shard below one round of wavefronts leaves SIMDs idle, so the series "
                      'tops out near 3.2 x at N = 8 by construction -- a SCALE curve that shows that is the model, not a defect; a weak N = 8 value below 0.90 x 8 x N1, or a '
                      'strong N = 8 speedup below 2.8, would be a finding.')
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('weak', 'strong', 'config4')}, indent=1))


if __name__ == '__main__':
    main()

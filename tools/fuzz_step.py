#!/usr/bin/env python3
"""Randomised cross-check of the single-step entry point (uvs_rmckf_step_f64: state X, P in HBM, the route Experiment.run() takes with an
external robot) against the per-row numpy oracle seeded with the same state: sequences of a few steps on random filters, every estimator,
shapes (8,6) / (6,6) / (2,6), annealing, thresholds / caps, large innovations (zero and subnormal MCKF weights), rank-deficient Jacobians
(numpy's pinv cutoff, which this kernel applies inline).   usage (GPU box): python tools/fuzz_step.py [cases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import uvs_amd as uvs  # noqa: E402
from oracle import rmckf_block  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
    bad, worst, n_steps = [], {'X': 0.0, 'P': 0.0, 'dq': 0.0}, 0
    t0 = time.time()
    for case in range(cases):
        method = ['GMCKF', 'KF', 'IMCCKF', 'MCKF'][case % 4]
        m = int(rng.choice([8, 8, 6, 2]))
        T, S = int(rng.integers(1, 40)), int(rng.integers(1, 8))
        bw = float(rng.choice([1.0, 10.0, 50.0]))
        anneal = bool(rng.random() < 0.4)
        gain = float(rng.uniform(0.05, 0.6))
        thr, cap = float(rng.choice([0.1, 1e-2, 1e-5])), int(rng.choice([1, 2, 5, 1000]))
        desired = 128 + 10 * rng.standard_normal(m)
        fp = uvs.engine.make_params(m, 6, method, bw, anneal, 0.05, 15.0, gain, desired, False, 0, 0, thr, cap)
        J = rng.standard_normal((T, m, 6)) * 50
        deficient = rng.random() < 0.15
        if deficient:                                            # a rank-deficient Jacobian estimate in filter 0: numpy's truncated solve
            J[0, :, 5] = J[0, :, 4] * 2.0
        x0 = J.reshape(T, m * 6)
        bank = uvs.engine.FilterBank(fp, T, x0)
        filt = [rmckf_block.BlockFilter(m, 6, x0[t], method, bw, anneal, 300, thr, cap) for t in range(T)]
        f_old = 128 + 20 * rng.standard_normal((T, m))
        dq = np.zeros((T, 6))
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device='cuda')      # noqa: E731
        scale = float(rng.choice([1.0, 30.0, 400.0]))            # 400: innovations of 38 sigma at sigma = 10 (zero / subnormal weights)
        for k in range(S):
            f = f_old + np.einsum('tmn,tn->tm', J, dq) * 0.05 + scale * rng.standard_t(2.0, size=(T, m))
            dq_dev, err_dev, kap_dev, st = bank.step(dev(f), dev(f_old), dev(dq), k)
            Xg, Pg = bank.X.cpu().numpy(), bank.P.cpu().numpy().reshape(T, m, 6, 6)
            cmd, status = dq_dev.cpu().numpy(), st.cpu().numpy()
            n_steps += T
            new_dq = np.zeros((T, 6))
            for t in range(T):
                with np.errstate(all='ignore'):
                    kappa = filt[t].step(f[t] - f_old[t], dq[t], k)
                    finite = np.all(np.isfinite(filt[t].X))
                tag = (case, method, m, T, k, t, bw, anneal, thr, cap, scale)
                if finite != (status[t] == 0):
                    bad.append(('status', tag, int(status[t]), bool(finite)))
                    continue
                if not finite:
                    filt[t] = rmckf_block.BlockFilter(m, 6, x0[t], method, bw, anneal, 300, thr, cap)     # restart this filter on both sides
                    filt[t].first = False
                    bank.X[t] = dev(x0[t])
                    bank.P[t] = torch.eye(6, dtype=torch.float64, device='cuda').repeat(m, 1, 1)
                    continue
                ref_cmd = rmckf_block.control_law(filt[t].X, f[t] - desired, kappa, gain)
                d = {'X': rel(Xg[t], filt[t].X.ravel()), 'P': rel(Pg[t], filt[t].P), 'dq': rel(cmd[t], ref_cmd)}
                # (after its first update the dependent columns of filter 0 are separated by rounding only: sigma_6 sits AT pinv's 1e-15 cutoff and
                # whether it is dropped is decided by the last bits of two different SVDs -- the command is compared on the exact deficiency of step 0 only)
                if deficient and t == 0 and k > 0:
                    d['dq'] = 0.0
                for key in d:
                    worst[key] = max(worst[key], d[key])
                if d['X'] > 1e-10 or d['P'] > 1e-9 or d['dq'] > 1e-7:
                    bad.append(('deviation', tag, d))
                new_dq[t] = ref_cmd
            # both sides continue from the ORACLE's state and command: deviations do not compound, every step is a fresh comparison
            bank.X.copy_(dev(np.stack([fl.X.ravel() for fl in filt])))
            bank.P.copy_(dev(np.stack([fl.P for fl in filt])).reshape(bank.P.shape))
            f_old, dq = f, np.clip(new_dq, -5, 5)
        if case % 50 == 49:
            print(f'{case + 1} cases, {n_steps} filter-steps, worst X {worst["X"]:.1e} P {worst["P"]:.1e} dq {worst["dq"]:.1e}, {len(bad)} mismatches, {time.time() - t0:.0f} s', flush=True)
    for b in bad[:30]:
        print('MISMATCH', b)
    print('done:', cases, 'cases,', n_steps, 'filter-steps,', len(bad), 'mismatches, worst', worst)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

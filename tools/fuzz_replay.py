#!/usr/bin/env python3
"""Long randomised cross-check of the REPLAY kernels (north_star's own I/O: recorded f / dq streams in, X / err / commanded dq out) against
oracle/c's uvs_oracle_replay: estimator, lane variant (library default mappings: row-group wavefronts, + control wavefronts, record path;
two-lane kernel; generic template), batch size, horizon, output layout, which streams are requested, bandwidth, annealing, MCKF threshold / cap,
noise scale.  Open loop: no feedback, so every trial is held to X <= 1e-9 and the command <= 1e-7; a trial whose streams turn X non-finite
must stop at the same step.   usage (GPU box): python tools/fuzz_replay.py [cases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import uvs_amd as uvs  # noqa: E402
from oracle import c_oracle  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
    worst_x, worst_c, bad, n_all, n_fail, n_multi = 0.0, 0.0, [], 0, 0, 0
    t0 = time.time()
    for case in range(cases):
        method = ['GMCKF', 'KF', 'IMCCKF', 'MCKF'][case % 4]
        m = int(rng.choice([8, 8, 8, 6, 2]))
        lane = int(rng.choice([0, 2, 4, -2, -4])) if m == 8 else int(rng.choice({6: [0, 2, 1], 2: [0, 1]}[m]))
        T = int(rng.choice([1, 15, 16, 17, 48, 63, 64, 65, 130, 200]))
        K = int(rng.integers(1, 100))
        bw = float(rng.choice([1.0, 10.0, 50.0]))
        anneal = bool(rng.random() < 0.4)
        gain = float(rng.uniform(0.05, 0.6))
        thr, cap = float(rng.choice([0.1, 1e-2, 1e-4])), int(rng.choice([1, 2, 5, 1000]))
        want_cmd = bool(rng.random() < 0.6)
        layout = 'ktc' if (m == 8 and rng.random() < 0.3) else 'kct'
        scale = float(rng.choice([0.3, 3.0, 30.0]))
        # consistent synthetic streams f_{k+1} = f_k + J dq_k dt + noise, heavy-tailed noise, one trial with an infinite feature
        J = rng.standard_normal((T, m, 6)) * 50
        dq = rng.standard_normal((T, K, 6)) * 0.2
        f = np.empty((T, K + 1, m))
        f[:, 0] = 128 + 20 * rng.standard_normal((T, m))
        for k in range(K):
            f[:, k + 1] = f[:, k] + np.einsum('tmn,tn->tm', J, dq[:, k]) * 0.05 + scale * rng.standard_t(2.0, size=(T, m))
        if K > 5 and T > 3 and rng.random() < 0.3:
            f[T // 2, K // 2, 0] = np.inf
        x0 = (J + 5 * rng.standard_normal((T, m, 6))).reshape(T, m * 6)
        desired = 128 + 10 * rng.standard_normal(m)
        ref = c_oracle.replay_batch(f, dq, x0, desired, method, bw, anneal, 300, gain, thr, cap)
        fp = uvs.engine.make_params(m, 6, method, bw, anneal, 0.05, 15.0, gain, desired, False, lane, K, thr, cap)
        to_dev = lambda a: torch.as_tensor(np.ascontiguousarray(a.transpose(1, 2, 0) if layout == 'kct' else a.transpose(1, 0, 2)), device='cuda')      # noqa: E731
        want = ('x', 'err', 'dqcmd') if want_cmd else ('x', 'err')
        out = uvs.engine.replay(fp, to_dev(f), to_dev(dq), torch.as_tensor(x0, device='cuda'), want=want, layout=layout)
        st, kd = out['status'].cpu().numpy(), out['k_done'].cpu().numpy()
        X = uvs.engine.as_tkc(out['x'], layout).cpu().numpy()
        cmd = uvs.engine.as_tkc(out['dqcmd'], layout).cpu().numpy() if want_cmd else None
        tag = (case, m, method, lane, T, K, bw, anneal, thr, cap, want_cmd, layout, scale)
        n_fail += int((ref['status'] == 1).sum())
        n_multi += int((ref['fpi'] >= 2).sum())
        for t in range(T):
            n_all += 1
            k1 = int(ref['k_done'][t])
            if st[t] != ref['status'][t] or kd[t] != k1:
                bad.append(('status', tag, t, int(st[t]), int(kd[t]), int(ref['status'][t]), k1))
                continue
            if k1:
                dx = rel(X[t, :k1], ref['X'][t, :k1])
                worst_x = max(worst_x, dx)
                dc = rel(cmd[t, :k1], ref['dq_cmd'][t, :k1]) if want_cmd else 0.0
                worst_c = max(worst_c, dc)
                if dx > 1e-9 or dc > 1e-7:
                    bad.append(('deviation', tag, t, dx, dc))
        if case % 50 == 49:
            print(f'{case + 1} cases, {n_all} trials ({n_fail} FAIL in the oracle, {n_multi} multi-pass MCKF steps), worst X {worst_x:.2e}, worst command {worst_c:.2e}, '
                  f'{len(bad)} mismatches, {time.time() - t0:.0f} s', flush=True)
    for b in bad[:40]:
        print('MISMATCH', b)
    print('done:', cases, 'cases,', n_all, 'trials,', len(bad), 'mismatches, worst X', f'{worst_x:.3e}', 'worst command', f'{worst_c:.3e}')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

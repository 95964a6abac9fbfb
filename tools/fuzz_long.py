#!/usr/bin/env python3
"""Long randomised cross-check of the closed-loop kernels against oracle/c (not part of the test suite: minutes of GPU and CPU time).
Every case draws estimator, lane variant, batch size, horizon, step, gain, bandwidth, annealing, noise law and scale, MCKF threshold and
cap; trials the oracle itself does not reproduce from starts moved by +1e-14 and -1e-14 (chaotic closed loops) are held to nothing, everybody else to
status / k_done exactly and trajectories <= 1e-8.   usage (GPU box): python tools/fuzz_long.py [cases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import uvs_amd as uvs  # noqa: E402
from oracle import c_oracle  # noqa: E402
import bench  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
    cfg = bench.config2()
    desired = cfg['experiments']['desired_f']
    lanes_of = {'GMCKF': [0, 2, 4, 1, -2, 8], 'KF': [0, 2, 4, -4, 8], 'IMCCKF': [0, 2, 4, -2, 8], 'MCKF': [0, 2, 4, -2, 1, -4]}
    shapes = {}
    for m in (8, 6, 2):                                            # 4, 3 and 1 features (BASELINE configs 2 and 1; the reference's tests/*_3_features)
        des = desired[:m]
        host = uvs.SyntheticPlant.ur10(des)
        pl = c_oracle.ur10_plant()
        pl.n_points = m // 2
        for i, w in enumerate(host.points):
            for c in range(3):
                pl.points[i][c] = w[c]
        shapes[m] = (des, host.to_struct(), pl)
    lin = uvs.LinearPlant.random(32, 7, seed=2)                   # the wide stress shape (BASELINE config 5): consistent linear plant, X0 supplied
    lrng = np.random.default_rng(5)
    q_goal = lin.q0 + lrng.uniform(-0.3, 0.3, 7)
    wide = dict(desired=lin.features(q_goal), q_goal=q_goal, x0=(lin.J * (1 + 0.1 * lrng.normal(size=lin.J.shape))).ravel(),
                struct=lin.to_struct('cuda'), pl=c_oracle.linear_plant(lin.J, lin.f0, lin.q0))
    worst, bad, n_calm, n_all, n_fail, n_multi = 0.0, [], 0, 0, 0, 0
    only = int(os.environ['UVS_FUZZ_ONLY']) if os.environ.get('UVS_FUZZ_ONLY') else None
    t0 = time.time()
    for case in range(cases):
        method = ['GMCKF', 'KF', 'IMCCKF', 'MCKF'][case % 4]
        m = int(rng.choice([8, 8, 8, 6, 2, 32]))
        n = 7 if m == 32 else 6
        if m == 32:
            desired, plant, pl = wide['desired'], wide['struct'], wide['pl']
            lane = int(rng.choice([0, 8, 16, 32, -8, -16])) if method != 'MCKF' else int(rng.choice([16, -16, 32]))      # MCKF: generic template only
        else:
            desired, plant, pl = shapes[m]
            lane = int(rng.choice(lanes_of[method])) if m == 8 else int(rng.choice({6: [0, 2, 1], 2: [0, 1]}[m]))
        layout = 'ktc' if (m == 32 and rng.random() < 0.6) else 'kct'
        T = int(rng.integers(1, 200)) if m != 32 else int(rng.choice([1, 7, 8, 9, 40, 64, 77]))
        K = int(rng.integers(1, 120))
        dt = float(rng.choice([0.02, 0.05, 0.1]))
        t_max = dt * (K + 1) + (dt / 2 if rng.random() < 0.5 else 5.0)
        gain = float(rng.uniform(0.05, 0.6))
        bw = float(rng.choice([1.0, 2.0, 10.0, 50.0]))
        anneal = bool(rng.random() < 0.4)
        law = rng.choice(['t2.5', 't1.2', 'cauchy', 'normal', 'none'])
        scale = float(rng.choice([0.5, 3.0, 10.0]))
        noise = {'t2.5': lambda: rng.standard_t(2.5, size=(T, K, m)), 't1.2': lambda: rng.standard_t(1.2, size=(T, K, m)),
                 'cauchy': lambda: rng.standard_cauchy(size=(T, K, m)), 'normal': lambda: rng.standard_normal((T, K, m)),
                 'none': lambda: np.zeros((T, K, m))}[law]() * scale
        thr, cap = float(rng.choice([0.1, 1e-2, 1e-4])), int(rng.choice([1, 2, 5, 1000]))
        if m == 32:
            q0 = wide['q_goal'] + rng.uniform(-0.15, 0.15, (T, 7))
            scale *= 0.3
            noise *= 0.3
        else:
            q0 = np.tile(cfg['experiments']['q_start'], (T, 1)).astype(float)
            q0[:, :3] += rng.uniform(-0.15, 0.15, (T, 3))
        if K > 4 and T > 2 and rng.random() < 0.15:               # a non-finite sample somewhere: FAIL (or, for MCKF and inf, a skipped correction)
            noise[T // 2, K // 2, 0] = np.inf if rng.random() < 0.7 else np.nan
        kw = dict(method=method, kernel_bw=bw, annealing=anneal, dt=dt, t_max=t_max, gain=gain, steps=K, want_x=True, fpi_threshold=thr, fpi_epoch_max=cap, plant=pl, x0=(wide['x0'] if m == 32 else None))
        fp = uvs.engine.make_params(m, n, method, bw, anneal, dt, t_max, gain, desired, m != 32, lane, K, thr, cap)
        # round 4's launch options: MCKF trials cut into 1-16 segments (tuned two-lane kernel only; others ignore it), the latency mapping,
        # strict pinv on one case in eight
        opts = int(rng.integers(1, 17)) << 8 if (method == 'MCKF' and rng.random() < 0.7) else 0
        if method == 'GMCKF' and m == 8 and lane == 2 and rng.random() < 0.5:      # RMCKF's segmented instantiation (two lanes per filter only)
            opts = int(rng.integers(2, 17)) << 8
        if lane == 0 and rng.random() < 0.4:
            opts |= 2
        if rng.random() < 0.125:
            opts |= 1
        if (opts >> 8) > 1 and rng.random() < 0.05:                # round 5: a lost hand-over now and then (the later segment recomputes after its spin budget)
            opts |= 4
        fp.reserved = opts
        # round 5: per-trial records for the X stream alone (the XREC instantiations of the (8,6) two-lane KF / IMCC-KF kernels; any other kernel takes the strides)
        x_layout = 'ktc' if (m == 8 and layout == 'kct' and rng.random() < 0.3) else None
        if only is not None and case != only:                      # UVS_FUZZ_ONLY=<case>: every draw above was made, nothing is run
            continue
        ref = c_oracle.closed_loop_batch(q0, noise, desired, **kw)
        ref2 = c_oracle.closed_loop_batch(q0 * (1.0 + 1e-14), noise, desired, **kw)
        ref3 = c_oracle.closed_loop_batch(q0 * (1.0 - 1e-14), noise, desired, **kw)   # (a single probe can land close by luck: case 6619 of seed 20261007)
        nz_dev = torch.as_tensor(np.ascontiguousarray(noise.transpose(1, 2, 0) if layout == 'kct' else noise.transpose(1, 0, 2)), device='cuda')
        x0_dev = torch.as_tensor(np.tile(wide['x0'], (T, 1)), device='cuda') if m == 32 else None
        out = uvs.engine.closed_loop(fp, plant, torch.as_tensor(q0, device='cuda'), nz_dev, x0_dev, want=('x', 'err', 'q'), layout=layout, x_layout=x_layout)
        st, kd = out['status'].cpu().numpy(), out['k_done'].cpu().numpy()
        X, E, Q = (uvs.engine.as_tkc(out[k], (x_layout or layout) if k == 'x' else layout).cpu().numpy() for k in ('x', 'err', 'q'))
        tag = (case, m, layout, x_layout, method, lane, T, K, dt, round(gain, 3), bw, anneal, law, scale, thr, cap, opts)
        n_fail += int((ref['status'] == 1).sum())
        if method == 'MCKF':
            n_multi += int((ref['fpi'] >= 2).sum())
        for t in range(T):
            n_all += 1
            k1 = int(ref['k_done'][t])
            calm = all(ref['status'][t] == o['status'][t] and k1 == int(o['k_done'][t]) and
                       (k1 == 0 or max(rel(o[r][t, :k1], ref[r][t, :k1]) for r in ('err', 'q', 'X')) <= 1e-11) for o in (ref2, ref3))
            if not calm:
                continue
            n_calm += 1
            if st[t] != ref['status'][t] or kd[t] != k1:
                bad.append(('status', tag, t, int(st[t]), int(kd[t]), int(ref['status'][t]), k1))
                continue
            if k1:
                d = max(rel(E[t, :k1], ref['err'][t, :k1]), rel(Q[t, :k1], ref['q'][t, :k1]), rel(X[t, :k1], ref['X'][t, :k1]))
                worst = max(worst, d)
                if d > 1e-8:
                    bad.append(('deviation', tag, t, d))
                    if only is not None:                           # where does it start, and does the oracle reproduce ITSELF from other starts?
                        dev = np.abs(X[t, :k1] - ref['X'][t, :k1]).max(axis=1) / max(np.abs(ref['X'][t, :k1]).max(), 1e-300)
                        first = int(np.argmax(dev > 1e-12)) if (dev > 1e-12).any() else -1
                        print(f'  trial {t}: first step with X deviation > 1e-12: {first}; deviation there {dev[first]:.3e}, at the end {dev[-1]:.3e}; oracle passes around it: '
                              f'{ref["fpi"][t, max(first - 3, 0):first + 4].tolist()}')
                        for eps in (1e-14, -1e-14, 3e-14, 1e-13, 1e-12):
                            r3 = c_oracle.closed_loop_batch(q0[t:t + 1] * (1.0 + eps), noise[t:t + 1], desired, **kw)
                            print(f'  oracle from the start moved by {eps:+.0e}: X deviates {rel(r3["X"][0, :k1], ref["X"][t, :k1]):.3e} from the oracle itself, passes there '
                                  f'{r3["fpi"][0, max(first - 3, 0):first + 4].tolist()}')
        if case % 25 == 24:
            print(f'{case + 1} cases, {n_all} trials ({n_calm} calm, {n_fail} FAIL in the oracle, {n_multi} multi-pass MCKF steps), worst calm deviation {worst:.2e}, '
                  f'{len(bad)} mismatches, {time.time() - t0:.0f} s', flush=True)
    for b in bad[:40]:
        print('MISMATCH', b)
    print('done:', cases, 'cases,', n_all, 'trials,', n_calm, 'calm,', len(bad), 'mismatches, worst', f'{worst:.3e}')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

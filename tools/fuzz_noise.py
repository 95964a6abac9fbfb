#!/usr/bin/env python3
"""Randomised cross-check of the on-device noise generator (uvs_noise_generate_f64 + uvs_pcg64_seed_u64) against the host restatement of the
reference's NoiseProfiler (uvs_amd.noise.noise_batch, itself bit-exact against 16 fixtures of the unmodified noise.py): noise type, its
parameters (alpha, beta, gamma, delta / std, mean, rho), feature count, hold on / off with random hold lengths, seeds up to 2^62, layouts.
Uniform / normal / mixture streams must agree to the bit (normals: to 2 ulp = 4.5e-16 on the rare wedge / tail samples, whose log1p / exp come from a different libm), the transcendental
alpha-stable branches to 2e-13.   usage (GPU box): python tools/fuzz_noise.py [cases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import uvs_amd as uvs  # noqa: E402


def ulp_study(samples=100_000_000):
    """`python tools/fuzz_noise.py --ulp [samples per alpha]`: ulp distance between the device generator and the host restatement of
    noise.py (numpy scalar arithmetic, bit-exact against the reference fixtures) over >= 1e8 alpha-stable samples per alpha, for the default
    kernels (Chambers-Mallows-Stuck powers folded into one exponential) and the as-written variant (UVS_NOISE_OPT_AS_WRITTEN), and what each
    costs on one sweep cell (65 536 trials: the T + 70 shared streams x 299 steps)."""
    import torch
    NT = uvs.NoiseType
    K, m = 299, 8
    T = max(64, samples // (K * m))
    for alpha in (1.0909090909090908, 1.5, 1.9090909090909092):
        params = dict(alpha=alpha, beta=0.0, gamma=1.0, delta=0.0)
        worst = {False: [], True: []}
        done = 0
        for lo in range(0, T, 4096):                                         # host generation in blocks (noise_batch: ~1.5 M samples/s per core)
            seeds = 123456 + np.arange(lo, min(T, lo + 4096), dtype=np.int64) * 80      # disjoint generator seeds (seed + 10 i, i < 8)
            host = uvs.noise_batch(NT.ALPHA_STABLE, params, seeds, m, K)
            hi = host.view(np.int64)
            for aw in (False, True):
                dev = uvs.engine.as_tkc(uvs.noise_device.generate(NT.ALPHA_STABLE, params, seeds, m, K, as_written=aw), 'kct').cpu().numpy()
                d = np.abs(np.ascontiguousarray(dev).view(np.int64) - hi)        # same sign (checked below): distance in ulp
                assert np.array_equal(np.sign(dev), np.sign(host))
                worst[aw].append((int(d.max()), float((d == 0).mean()), float((d <= 1).mean()), float((d <= 2).mean()), float(np.quantile(d, 0.9999)),
                                  float(np.abs(host.ravel()[np.argmax(d)]))))
            done += host.size
        for aw in (False, True):
            w = np.array(worst[aw])
            print(f'alpha {alpha:.4f} {"as written" if aw else "folded    "}: {done / 1e6:.0f} M samples, max {int(w[:, 0].max())} ulp (at |x| = {w[np.argmax(w[:, 0]), 5]:.3g}), '
                  f'99.99 % <= {w[:, 4].max():.0f} ulp, exact {w[:, 1].mean():.4f}, <= 1 ulp {w[:, 2].mean():.4f}, <= 2 ulp {w[:, 3].mean():.4f}', flush=True)
    # cost of one sweep cell's noise (the shared T + 70 streams) and of the dense per-trial generation
    params = dict(alpha=1.5, beta=0.0, gamma=1.0, delta=0.0)
    for label, fn in (('shared streams (65 536 + 70) x 299', lambda aw: uvs.noise_device.generate_shared(NT.ALPHA_STABLE, params, 123456, 65536, m, K, as_written=aw)),
                      ('dense 65 536 x 8 x 299', lambda aw: uvs.noise_device.generate(NT.ALPHA_STABLE, params, 123456 + np.arange(65536), m, K, as_written=aw))):
        for aw in (False, True):
            for _ in range(3):
                fn(aw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn(aw)
            e1.record()
            torch.cuda.synchronize()
            print(f'{label}, {"as written" if aw else "folded    "}: {e0.elapsed_time(e1) / 10:.3f} ms per call (seeding included)', flush=True)


def numpy_own_error(n=4_000_000):
    """`python tools/fuzz_noise.py --truth` (CPU only): how far numpy's OWN float64 evaluation of noise.py:188-191 is from the exact value of that formula for
    the same (V, W) -- the formula evaluated in 80-bit long double.  The roundings of the intermediate arguments (alpha V, V (1 - alpha), the quotient under the
    power) are amplified by the functions around them: that, not libm quality, sets the scale on which "ulp from numpy" has to be read."""
    rng = np.random.Generator(np.random.PCG64(5))
    L = np.longdouble
    for alpha in (1.0909090909090908, 1.5, 1.9090909090909092):
        V = rng.uniform(-np.pi / 2, np.pi / 2, n)
        W = -np.log(rng.uniform(0, 1, n))
        x = (np.sin(alpha * V) / (np.cos(V) ** (1 / alpha))) * (np.cos(V * (1 - alpha)) / W) ** ((1 - alpha) / alpha)
        a, Vl, Wl = L(alpha), V.astype(L), W.astype(L)
        xt = (np.sin(a * Vl) / (np.cos(Vl) ** (1 / a))) * (np.cos(Vl * (1 - a)) / Wl) ** ((1 - a) / a)
        err = (np.abs(x.astype(L) - xt) / np.spacing(np.abs(x)).astype(L)).astype(float)
        print(f'alpha {alpha:.4f}: numpy float64 against the 80-bit evaluation of the same formula, {n / 1e6:.0f} M samples: max {err.max():.1f} ulp, '
              f'99.99 % <= {np.quantile(err, 0.9999):.1f} ulp, mean {err.mean():.2f} ulp')


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--truth':
        return numpy_own_error()
    if len(sys.argv) > 1 and sys.argv[1] == '--ulp':
        return ulp_study(int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    NT = uvs.NoiseType
    bad, worst, n, n_shared = [], 0.0, 0, 0
    t0 = time.time()
    for case in range(cases):
        kind = [NT.WHITE_NOISE, NT.GAUSSIAN_MIXTURE, NT.GAUSSIAN_BIMODAL, NT.ALPHA_STABLE, NT.UNIFORM][case % 5]
        m = int(rng.choice([2, 4, 8, 8, 32]))
        T = int(rng.integers(1, 150))
        K = int(rng.integers(1, 120))
        hold = bool(rng.random() < 0.4) and kind != NT.UNIFORM
        hold_cnt = int(rng.integers(1, 15))
        if kind == NT.ALPHA_STABLE:
            alpha = float(rng.choice([2.0, 1.0, 0.5, 1.5, 1.2, 1.0909090909090908, 0.8, 1.9]))
            if rng.random() < 0.4:                               # any index: both sides of the launcher's error gate for the beta = 0 instantiation
                alpha = float(rng.uniform(0.05, 1.9999))
            beta = float(rng.choice([0.0, 0.0, 0.5, -0.5, 1.0])) if alpha != 0.5 else float(rng.choice([1.0, -1.0, 0.0]))
            params = dict(alpha=alpha, beta=beta, gamma=float(rng.choice([1.0, 2.0, 0.5])), delta=float(rng.choice([0.0, 1.0, -3.0])))
        elif kind == NT.UNIFORM:
            params = {}
        else:
            params = dict(std=float(rng.choice([1.0, 2.0, 0.1])), mean=float(rng.choice([50.0, 30.0, 5.0])), rho=float(rng.choice([0.1, 0.3, 0.0, 1.0])))
        seeds = (rng.integers(0, 2 ** 62, T) if rng.random() < 0.3 else rng.integers(0, 10 ** 6) + np.arange(T)).astype(np.int64)
        layout = str(rng.choice(['kct', 'ktc']))
        host = uvs.noise_batch(kind, params, seeds, m, K, hold, hold_cnt)
        dev = uvs.engine.as_tkc(uvs.noise_device.generate(kind, params, seeds, m, K, hold, hold_cnt, layout=layout, device='cuda'), layout).cpu().numpy()
        n += host.size
        tag = (case, kind.name, params, m, T, K, hold, hold_cnt, layout)
        if kind == NT.ALPHA_STABLE:                                  # the as-written variant: the same gate (it is tighter in fact: --ulp measures it)
            aw = uvs.engine.as_tkc(uvs.noise_device.generate(kind, params, seeds, m, K, hold, hold_cnt, layout=layout, device='cuda', as_written=True), layout).cpu().numpy()
            fin_aw = np.isfinite(host)
            if not np.array_equal(np.isfinite(aw), fin_aw) or np.any(np.abs(aw[fin_aw] - host[fin_aw]) > 2e-13 * np.abs(host[fin_aw]) + 1e-12 * (1 + abs(params.get('delta', 0.0)))):
                bad.append(('as-written variant', tag))
        # round 5: where the trials' streams alias (consecutive seeds, one generator per feature, no hold) the shared-stream generator -- T + 10 (m - 1)
        # streams once, in chunks behind a PCG64 jump -- must return the per-trial generator's bits
        if uvs.noise_device.shares_streams(kind, hold, seeds):
            _, view = uvs.noise_device.generate_shared(kind, params, int(seeds[0]), T, m, K)
            same = np.array_equal(view.permute(2, 0, 1).cpu().numpy().view(np.int64), np.ascontiguousarray(dev).view(np.int64))
            n_shared += 1
            if not same:
                bad.append(('shared streams differ from per-trial streams', tag))
                continue
        if not np.all(np.isfinite(host) == np.isfinite(dev)):
            bad.append(('finite pattern', tag))
            continue
        fin = np.isfinite(host)
        exact_kind = kind in (NT.WHITE_NOISE, NT.GAUSSIAN_MIXTURE, NT.GAUSSIAN_BIMODAL, NT.UNIFORM) or (kind == NT.ALPHA_STABLE and params['alpha'] == 2.0)
        scale = np.maximum(np.abs(host[fin]), 1e-300)
        d = np.abs(dev[fin] - host[fin]) / scale
        if exact_kind:
            if np.mean(dev[fin] != host[fin]) > 2e-3 or (len(d) and d.max() > 4.5e-16):
                bad.append(('bits', tag, float(np.mean(dev[fin] != host[fin])), float(d.max())))
        else:
            atol = 1e-12 * (1 + abs(params.get('delta', 0.0)))                          # gamma x + delta cancels near 0
            viol = np.abs(dev[fin] - host[fin]) > 2e-13 * np.abs(host[fin]) + atol
            if viol.any():
                bad.append(('tolerance', tag, float(d.max())))
            worst = max(worst, float(np.median(d)) if len(d) else 0.0)
        if case % 50 == 49:
            print(f'{case + 1} cases ({n_shared} also through the shared-stream generator), {n / 1e6:.1f} M samples, {len(bad)} mismatches, {time.time() - t0:.0f} s', flush=True)
    for b in bad[:30]:
        print('MISMATCH', b)
    print('done:', cases, 'cases,', f'{n / 1e6:.1f} M samples,', len(bad), 'mismatches')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

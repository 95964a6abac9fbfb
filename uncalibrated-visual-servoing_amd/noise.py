"""Measurement-noise streams with the reference's interface (noise.py) plus a batched generator.

``NoiseType`` / ``NoiseProfiler`` keep the reference's names, constructor signature, seeding rule and
``getNoise()`` semantics (noise.py:7-118) so existing callers work unchanged.  The Monte-Carlo driver does
not call ``getNoise()`` 20 million times; it uses ``noise_batch`` which produces the identical streams for
many trials at once with vectorised numpy on the same ``numpy.random.Generator(PCG64)`` bit streams the
reference consumes (the reference's own third-party RNG; seeds ``seed + 10*i`` and ``2*seed + i``,
noise.py:59,70).  Uniform / normal / Cauchy / mixture draws are bit-identical to the scalar path; the
Chambers-Mallows-Stuck transform goes through numpy's array sin/cos/power, which may differ from its scalar
path in the last bit.
"""
import logging
from enum import Enum

import numpy as np
from numpy.random import PCG64, Generator, default_rng

OUTLIER_ABS = 20          # noise.py:103


class NoiseType(Enum):
    WHITE_NOISE = 1
    GAUSSIAN_MIXTURE = 2
    GAUSSIAN_BIMODAL = 3
    ALPHA_STABLE = 4
    UNIFORM = 5

    @classmethod
    def numberOfGenerators(cls, type) -> int:
        return {cls.WHITE_NOISE: 1, cls.GAUSSIAN_MIXTURE: 2, cls.GAUSSIAN_BIMODAL: 3, cls.ALPHA_STABLE: 1, cls.UNIFORM: 1}.get(type, 0)


def _unpack(params):
    return params['noise_params'] if 'noise_params' in params else params


def _cms(alpha, beta, V, W):
    """Chambers-Mallows-Stuck general branches (noise.py:188-199) on arrays or scalars."""
    if beta == 0:
        return (np.sin(alpha * V) / (np.cos(V) ** (1 / alpha))) * (np.cos(V * (1 - alpha)) / W) ** ((1 - alpha) / alpha)
    if alpha != 1:
        c = beta * np.tan(np.pi * alpha / 2)
        B = np.arctan(c)
        S = (1 + c ** 2) ** (1 / (2 * alpha))
        return S * np.sin(alpha * V + B) / (np.cos(V) ** (1 / alpha)) * (np.cos((1 - alpha) * V - B) / W) ** ((1 - alpha) / alpha)
    sv = np.pi / 2 + beta * V
    return 2 / np.pi * (sv * np.tan(V) - beta * np.log((W * np.cos(V)) / sv))


def _scale_shift(alpha, beta, gamma, delta, x):
    """noise.py:201-205."""
    if alpha == 1:
        return gamma * x + (2 / np.pi) * beta * gamma * np.log(gamma) + delta
    return gamma * x + delta


class NoiseProfiler:
    """Drop-in for the reference class (noise.py:29-118): one private generator per feature (and per mixture
    component), pair-wise hold of outliers, ``getNoise()`` returning the same buffer object on every call."""

    def __init__(self, num_features: int, noise_type, seed: int = None, logger: object = None, noise_hold: bool = False,
                 noise_hold_cnt: int = 0, **noise_params) -> None:
        self.num_features = num_features
        self.noise_type = noise_type
        self.seed = seed
        self.logger = logging.getLogger(__name__)
        if logger is not None:
            self.logger.setLevel(logger.level)
        p = _unpack(noise_params)
        self.params = dict(p)
        if noise_type in (NoiseType.WHITE_NOISE, NoiseType.GAUSSIAN_MIXTURE, NoiseType.GAUSSIAN_BIMODAL):
            self.std = p['std']
            if noise_type != NoiseType.WHITE_NOISE:
                self.mean, self.rho = p['mean'], p['rho']
                self.rhoGenerators = [default_rng() if seed is None else Generator(PCG64(2 * seed + i)) for i in range(num_features)]
        elif noise_type == NoiseType.ALPHA_STABLE:
            self.alpha, self.beta, self.gamma, self.delta = p['alpha'], p['beta'], p['gamma'], p['delta']
        self.generators = [default_rng() if seed is None else Generator(PCG64(seed + 10 * i))
                           for i in range(NoiseType.numberOfGenerators(noise_type) * num_features)]
        self.noise_hold = bool(noise_hold)
        self.noise_hold_cnt_max_predefined = noise_hold_cnt if noise_hold else 0
        self.noise_hold_cnt = np.zeros(num_features // 2)
        self.noise_hold_cnt_max = np.zeros(num_features // 2)
        self.noise = np.zeros(num_features)

    def _sample(self, idx):
        t, g, m = self.noise_type, self.generators, self.num_features
        if t == NoiseType.WHITE_NOISE:
            return g[idx].normal(loc=0.0, scale=self.std)
        if t == NoiseType.UNIFORM:
            return g[idx].uniform()
        if t in (NoiseType.GAUSSIAN_MIXTURE, NoiseType.GAUSSIAN_BIMODAL):
            u = self.rhoGenerators[idx].uniform(low=0, high=1)
            if u > self.rho:
                return g[idx].normal(loc=0.0, scale=self.std)
            if t == NoiseType.GAUSSIAN_MIXTURE or u > self.rho / 2:
                return g[idx + m].normal(loc=self.mean, scale=self.std)
            return g[idx + 2 * m].normal(loc=-self.mean, scale=self.std)
        a, b = self.alpha, self.beta
        if a == 2:
            x = g[idx].normal(loc=0.0, scale=np.sqrt(2))
        elif a == 1 and b == 0:
            x = np.tan(g[idx].uniform(low=-np.pi / 2, high=np.pi / 2))
        elif a == 0.5 and abs(b) == 1:
            x = b / (g[idx].normal(loc=0.0, scale=1.0) ** 2)
        else:
            V = g[idx].uniform(low=-np.pi / 2, high=np.pi / 2)
            W = -np.log(g[idx].uniform(low=0.0, high=1.0))
            x = _cms(a, b, V, W)
        return _scale_shift(a, b, self.gamma, self.delta, x)

    def getNoise(self):
        for i in range(self.num_features // 2):
            if self.noise_hold_cnt[i] >= self.noise_hold_cnt_max[i]:
                self.noise_hold_cnt[i] = 0
                a, b = self._sample(2 * i), self._sample(2 * i + 1)
                outlier = abs(a) > OUTLIER_ABS or abs(b) > OUTLIER_ABS
                self.noise_hold_cnt_max[i] = self.noise_hold_cnt_max_predefined if outlier else 0
                self.noise[2 * i], self.noise[2 * i + 1] = a, b
            else:
                self.noise_hold_cnt[i] += 1
        return self.noise


# ------------------------------------------------------------------------------------------------ batched streams
def _draws(noise_type, p, seed, m, K):
    """Un-held sample streams for one trial: (K, m) values in draw order (one row per *fresh* draw)."""
    gens = lambda off=0: [Generator(PCG64(seed + 10 * (i + off))) for i in range(m)]      # noqa: E731
    if noise_type == NoiseType.UNIFORM:
        return np.stack([g.random(K) for g in gens()], axis=1)
    if noise_type == NoiseType.WHITE_NOISE:
        return np.stack([g.normal(0.0, p['std'], K) for g in gens()], axis=1)
    if noise_type in (NoiseType.GAUSSIAN_MIXTURE, NoiseType.GAUSSIAN_BIMODAL):
        out = np.empty((K, m))
        for i in range(m):
            u = Generator(PCG64(2 * seed + i)).random(K)
            comp = np.where(u > p['rho'], 0, 1 if noise_type == NoiseType.GAUSSIAN_MIXTURE else np.where(u > p['rho'] / 2, 1, 2))
            col = np.empty(K)
            for c, (loc, off) in enumerate(((0.0, 0), (p['mean'], m), (-p['mean'], 2 * m))):
                sel = comp == c
                if sel.any():                                     # each component generator advances only when selected
                    col[sel] = Generator(PCG64(seed + 10 * (i + off))).normal(loc, p['std'], int(sel.sum()))
            out[:, i] = col
        return out
    a, b = p['alpha'], p['beta']
    if a == 2:
        x = np.stack([g.normal(0.0, np.sqrt(2), K) for g in gens()], axis=1)
    elif a == 1 and b == 0:
        x = np.tan(np.stack([-np.pi / 2 + np.pi * g.random(K) for g in gens()], axis=1))
    elif a == 0.5 and abs(b) == 1:
        x = b / np.stack([g.normal(0.0, 1.0, K) for g in gens()], axis=1) ** 2
    else:
        u = np.stack([g.random(2 * K).reshape(K, 2) for g in gens()], axis=1)             # (K, m, 2): V then W per draw
        x = _cms(a, b, -np.pi / 2 + np.pi * u[..., 0], -np.log(u[..., 1]))
    return _scale_shift(a, b, p['gamma'], p['delta'], x)


def _apply_hold(draws, hold_cnt):
    """Replay the pair-wise hold state machine (noise.py:82-116) over fresh-draw streams (K, m) -> per-call values."""
    K, m = draws.shape
    if hold_cnt <= 0:
        return draws
    out = np.empty_like(draws)
    for pair in range(m // 2):
        pos, cnt, cnt_max = 0, 0, 0
        cur = np.zeros(2)
        for k in range(K):
            if cnt >= cnt_max:
                cnt = 0
                cur = draws[pos, 2 * pair:2 * pair + 2]
                pos += 1
                cnt_max = hold_cnt if (abs(cur[0]) > OUTLIER_ABS or abs(cur[1]) > OUTLIER_ABS) else 0
            else:
                cnt += 1
            out[k, 2 * pair:2 * pair + 2] = cur
    return out


def noise_batch(noise_type, noise_params, seeds, num_features, steps, noise_hold=False, noise_hold_cnt=0, out=None):
    """Noise for many trials: ``out[t, k, :]`` equals the k-th ``getNoise()`` of
    ``NoiseProfiler(num_features, noise_type, seeds[t], noise_hold=..., noise_hold_cnt=..., **noise_params)``.
    ``out`` may be any array-like of shape (T, steps, num_features) (e.g. a transposed view of a trial-fastest buffer)."""
    p = _unpack(dict(noise_params)) if isinstance(noise_params, dict) else noise_params
    seeds = np.asarray(seeds).ravel()
    if out is None:
        out = np.empty((len(seeds), steps, num_features))
    hold = noise_hold_cnt if noise_hold else 0
    for t, seed in enumerate(seeds):
        out[t] = _apply_hold(_draws(noise_type, p, int(seed), num_features, steps), hold)
    return out

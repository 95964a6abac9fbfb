"""Oracle (test infrastructure): scalar restatement of the reference noise streams.

Follows ``noise.py`` of the reference call-for-call so that every
``numpy.random.Generator`` is consumed in the same order:

* generator seeding ``PCG64(seed + 10*i)`` for ``i < gens*m`` (noise.py:66-70),
  mixture selectors ``PCG64(2*seed + i)`` (noise.py:55-59);
* per feature *pair* hold state machine, outlier test ``abs(x) > 20``
  (noise.py:82-116);
* white (noise.py:125-128), Gaussian mixture (:130-138), bimodal (:140-150),
  uniform (:120-123) and Chambers-Mallows-Stuck alpha-stable draws with the
  three special cases and the scale/shift (:152-207).

Third-party arithmetic: ``numpy.random.Generator(PCG64)`` (numpy==2.2.4 pinned
by the reference's requirements.txt:1; 2.2.6 in the build image).  It is not
vendored in the reference; streams are pinned by ``tests/golden/noise_*.npz``.
"""
from numpy.random import Generator, PCG64
import numpy as np

WHITE_NOISE, GAUSSIAN_MIXTURE, GAUSSIAN_BIMODAL, ALPHA_STABLE, UNIFORM = 1, 2, 3, 4, 5
_GENS_PER_FEATURE = {WHITE_NOISE: 1, GAUSSIAN_MIXTURE: 2, GAUSSIAN_BIMODAL: 3, ALPHA_STABLE: 1, UNIFORM: 1}
OUTLIER_ABS = 20


class NoiseStreamRef:
    def __init__(self, m, kind, seed, hold=False, hold_cnt=0, **p):
        self.m, self.kind, self.p = m, kind, p
        self.gen = [Generator(PCG64(seed + 10 * i)) for i in range(_GENS_PER_FEATURE[kind] * m)]
        self.sel = [Generator(PCG64(2 * seed + i)) for i in range(m)] if kind in (GAUSSIAN_MIXTURE, GAUSSIAN_BIMODAL) else None
        self.hold_len = hold_cnt if hold else 0
        self.cnt = np.zeros(m // 2)
        self.cnt_max = np.zeros(m // 2)
        self.value = np.zeros(m)

    # one sample for feature idx; consumes generators exactly like noise.py
    def _draw(self, idx):
        p, m = self.p, self.m
        if self.kind == WHITE_NOISE:
            return self.gen[idx].normal(loc=0.0, scale=p['std'])
        if self.kind == UNIFORM:
            return self.gen[idx].uniform()
        if self.kind == GAUSSIAN_MIXTURE:
            u = self.sel[idx].uniform(low=0, high=1)
            if u > p['rho']:
                return self.gen[idx].normal(loc=0.0, scale=p['std'])
            return self.gen[idx + m].normal(loc=p['mean'], scale=p['std'])
        if self.kind == GAUSSIAN_BIMODAL:
            u = self.sel[idx].uniform(low=0, high=1)
            if u > p['rho']:
                return self.gen[idx].normal(loc=0.0, scale=p['std'])
            if u > p['rho'] / 2:
                return self.gen[idx + m].normal(loc=p['mean'], scale=p['std'])
            return self.gen[idx + 2 * m].normal(loc=-p['mean'], scale=p['std'])
        return self._alpha_stable(idx)

    def _alpha_stable(self, idx):
        a, b, gam, dlt = self.p['alpha'], self.p['beta'], self.p['gamma'], self.p['delta']
        g = self.gen[idx]
        if a == 2:
            x = g.normal(loc=0.0, scale=np.sqrt(2))
        elif a == 1 and b == 0:
            x = np.tan(g.uniform(low=-np.pi / 2, high=np.pi / 2))
        elif a == 0.5 and abs(b) == 1:
            x = b / (g.normal(loc=0.0, scale=1.0) ** 2)
        else:
            V = g.uniform(low=-np.pi / 2, high=np.pi / 2)
            W = -np.log(g.uniform(low=0.0, high=1.0))
            if b == 0:
                x = (np.sin(a * V) / (np.cos(V) ** (1 / a))) * (np.cos(V * (1 - a)) / W) ** ((1 - a) / a)
            elif a != 1:
                c = b * np.tan(np.pi * a / 2)
                B = np.arctan(c)
                S = (1 + c ** 2) ** (1 / (2 * a))
                x = S * np.sin(a * V + B) / (np.cos(V) ** (1 / a)) * (np.cos((1 - a) * V - B) / W) ** ((1 - a) / a)
            else:
                sv = np.pi / 2 + b * V
                x = 2 / np.pi * (sv * np.tan(V) - b * np.log((W * np.cos(V)) / sv))
        if a == 1:
            return gam * x + (2 / np.pi) * b * gam * np.log(gam) + dlt
        return gam * x + dlt

    def next(self):
        """One getNoise() call -> fresh copy of the m noise values (reference aliases its buffer)."""
        for pair in range(self.m // 2):
            if self.cnt[pair] >= self.cnt_max[pair]:
                self.cnt[pair] = 0
                a = self._draw(2 * pair)
                b = self._draw(2 * pair + 1)
                self.cnt_max[pair] = self.hold_len if (abs(a) > OUTLIER_ABS or abs(b) > OUTLIER_ABS) else 0
                self.value[2 * pair], self.value[2 * pair + 1] = a, b
            else:
                self.cnt[pair] += 1
        return self.value.copy()

    def take(self, steps):
        return np.stack([self.next() for _ in range(steps)])

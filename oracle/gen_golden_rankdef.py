"""Fixtures for numpy.linalg.pinv semantics on rank-deficient Jacobians (experiment.py:312: SVD, rcond = 1e-15).

BUILD-CONTAINER ONLY (imports /root/reference through gen_golden.py).  The reference reaches a rank-deficient X only with
``initial_guess=False``, where it draws X0 from an UNSEEDED ``np.random.default_rng().random`` (experiment.py:117).  To make it start
from a structured X0 without touching its code, ``numpy.random.default_rng`` is replaced -- for the duration of ``Experiment.run()``
only -- by a factory whose ``.random(shape)`` hands back the chosen X0; every other line is the reference's.  The subspace spanned by
the rows of X0 is invariant under the filter (the command lies in it, P stays a multiple of I plus terms inside it), so X keeps its
rank deficiency for the whole run and every control step goes through pinv's truncation.

    python oracle/gen_golden_rankdef.py       # writes tests/golden/rankdef_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G                                                # noqa: E402


class _FixedX0:
    def __init__(self, x0):
        self.x0 = x0

    def random(self, shape):
        return self.x0.reshape(shape).copy()


def cases():
    rng = np.random.default_rng(0)
    out = {}
    out['rank4_product'] = (rng.normal(size=(8, 4)) * 30) @ rng.normal(size=(4, 6))        # rank 4 up to rounding of the product
    x = rng.normal(size=(8, 6)) * 50
    x[:, 5] = x[:, 4]
    out['dup_col'] = x                                                # two identical columns: rank 5, deficiency decays (chaotic later)
    x = rng.normal(size=(8, 6)) * 50
    x[:, 2] = 0
    x[:, 3] = 2 * x[:, 1]
    out['zero_and_scaled_col'] = x                                    # a zero column AND a dependent one: rank 4
    out['rank1'] = np.outer(rng.normal(size=8), rng.normal(size=6)) * 50
    return out


def boundary_cases():
    """Jacobians ON the boundary of the kernels' rank watch: badly scaled AND nearly parallel columns.  numpy's pinv
    (experiment.py:312) decides by singular values; an unpivoted QR's diagonal does not show them.
      scaled_1e12_par_1em9  column 5 = 1e12 (column 4 + 1e-9 noise): sigma_min / sigma_max 8e-22 -> numpy truncates; |R_cc| spread 1.5e3
      scaled_1e6_par_1em12  column 5 = 1e6 (column 4 + 1e-12 noise): 5e-19 -> truncates; |R_cc| spread 2.8e6
      scaled_1e12_indep     column 5 scaled by 1e12, independent: 2e-13 -> NOT truncated (bad scaling alone is harmless)
      kahan_c1000           Q R with R = unit upper triangular, every off-diagonal entry -1000: 3e-19 -> truncates, every |R_ij| ordinary
    """
    rng = np.random.default_rng(20260)
    base = rng.normal(size=(8, 6)) * 50
    out = {}
    x = base.copy()
    x[:, 5] = 1e12 * (x[:, 4] + 1e-9 * 50 * rng.normal(size=8))
    out['scaled_1e12_par_1em9'] = x
    x = base.copy()
    x[:, 5] = 1e6 * (x[:, 4] + 1e-12 * 50 * rng.normal(size=8))
    out['scaled_1e6_par_1em12'] = x
    x = base.copy()
    x[:, 5] *= 1e12
    out['scaled_1e12_indep'] = x
    q, _ = np.linalg.qr(rng.normal(size=(8, 6)))
    out['kahan_c1000'] = 50.0 * q @ (np.eye(6) - 1000.0 * np.triu(np.ones((6, 6)), 1))
    return out


def main():
    AS = dict(alpha=1.5, beta=0, gamma=1, delta=0)
    for name, x0 in cases().items():
        real = np.random.default_rng
        np.random.default_rng = lambda *a, _x=x0, **k: _FixedX0(_x.ravel())
        try:
            for method in (G.E.Method.GMCKF, G.E.Method.KF):
                if method == G.E.Method.KF and name != 'rank4_product':
                    continue
                G.save_closed(f'{method.name.lower()}_{name}', method, G.NoiseType.ALPHA_STABLE, AS, 123456, prefix='rankdef_',
                              f_init=np.zeros(8), extra={'x0': x0.ravel()}, initial_guess=False)
        finally:
            np.random.default_rng = real


def main_boundary():
    AS = dict(alpha=1.5, beta=0, gamma=1, delta=0)
    for name, x0 in boundary_cases().items():
        sv = np.linalg.svd(x0, compute_uv=False)
        r = np.abs(np.linalg.qr(x0, mode='r'))
        real = np.random.default_rng
        np.random.default_rng = lambda *a, _x=x0, **k: _FixedX0(_x.ravel())
        try:
            G.save_closed(f'gmckf_{name}', G.E.Method.GMCKF, G.NoiseType.ALPHA_STABLE, AS, 123456, prefix='rankdef_', f_init=np.zeros(8),
                          extra={'x0': x0.ravel(), 'sv_ratio_x0': sv.min() / sv.max(), 'truncated_x0': int((sv <= 1e-15 * sv.max()).sum()),
                                 'diag_spread_x0': np.diag(r).max() / np.diag(r).min(), 'entry_spread_x0': r.max() / np.diag(r).min()},
                          initial_guess=False)
        finally:
            np.random.default_rng = real


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'boundary':
        main_boundary()                                               # only the new fixtures (the others stay byte-identical)
    else:
        main()
        main_boundary()

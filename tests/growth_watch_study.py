"""Analysis helper (test infrastructure): VERDICT r4 #5 asks for a solution-growth watch -- mark a control-law solve when
max|sol| * max|R_ij| > 2^k * max|Q^T b|.  This script evaluates that quantity (and the ratio of the plain least-squares command to numpy's pinv
command) on the reference fixtures: the Kahan-like one, the other rank-deficient ones, and every healthy closed-loop fixture.
`python tests/growth_watch_study.py`"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import golden_names, load_golden  # noqa: E402


def study(name):
    g = load_golden(name)
    X = g['X']                                                   # (steps, 48) at X_steps
    steps = g['X_steps']
    meth = g['meta']['method']
    out = []
    for i, k in enumerate(steps):
        J = X[i].reshape(8, 6)
        err = g['err'][k]
        if meth == 'GMCKF':
            kap = np.exp(-0.5 * g['e'][k] ** 2 / g['sigma'][k] ** 2)
        else:
            kap = np.ones(8)
        b = kap * err
        if not np.all(np.isfinite(J)):
            continue
        Q, R = np.linalg.qr(J)
        c = Q.T @ b
        with np.errstate(all='ignore'):
            try:
                sol = np.linalg.solve(R, c)
            except np.linalg.LinAlgError:
                continue
        pin = np.linalg.pinv(J) @ b
        growth = np.abs(sol).max() * np.abs(R).max() / max(np.abs(c).max(), 1e-300)
        sv = np.linalg.svd(J, compute_uv=False)
        dev = np.abs(sol - pin).max() / max(np.abs(pin).max(), 1e-300)
        spread = np.abs(R).max() / max(np.abs(np.diag(R)).min(), 1e-300)
        out.append((growth, dev, sv[0] / max(sv[-1], 1e-300), spread))
    return np.array(out)


if __name__ == '__main__':
    print(f'{"fixture":44s} {"max growth":>11s} {"p50 growth":>11s} {"max LS/pinv dev":>16s} {"max cond":>10s} {"max spread":>11s}  steps with dev > 1e-6: their min growth / min spread')
    for name in golden_names('rankdef_') + golden_names('closed_') + golden_names('fpi_'):
        a = study(name)
        if not len(a):
            continue
        bad = a[a[:, 1] > 1e-6]
        print(f'{name:44s} {a[:, 0].max():11.3g} {np.median(a[:, 0]):11.3g} {a[:, 1].max():16.3g} {a[:, 2].max():10.3g} {a[:, 3].max():11.3g}  '
              + (f'{len(bad)} steps: min growth {bad[:, 0].min():.3g}, min spread {bad[:, 3].min():.3g}' if len(bad) else '-'))

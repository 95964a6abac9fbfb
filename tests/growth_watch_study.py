"""Analysis helper (test infrastructure): VERDICT r4 #5 asks for a solution-growth watch -- mark a control-law solve when
max|sol| * max|R_ij| > 2^k * max|Q^T b|.  This script evaluates that quantity (and the ratio of the plain least-squares command to numpy's pinv
command) on the reference fixtures: the Kahan-like one, the other rank-deficient ones, and every healthy closed-loop fixture.
`python tests/growth_watch_study.py`

`python tests/growth_watch_study.py --wide` (round 6): the same question for the NORMAL-EQUATION solve of the wide (32,7) kernel, where squaring J
destroys its smallest singular value and the solution of a Kahan-like system barely grows; evaluated on block-oracle closed loops of BASELINE
config 5's plant: growth of the solution, spread of the squared Cholesky pivots, and the size of the refinement correction against the
solution -- the quantity rmckf_wide.hpp gates at 2^-20 (healthy: <= 7e-14; Kahan-like with cond >= 1e18: breakdown or >= 2e-3 on every step)."""
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


def study_wide():
    import scipy.linalg as sl
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import uvs_amd as uvs
    from oracle import rmckf_block
    plant = uvs.LinearPlant.random(32, 7, seed=2)
    rng = np.random.default_rng(5)
    q_goal = plant.q0 + rng.uniform(-0.3, 0.3, 7)
    des = plant.features(q_goal)

    def ratios(X, errs):
        out, broke = [], 0
        for J, y in zip(X.reshape(-1, 32, 7), errs):
            G = J.T @ J
            try:
                c = sl.cho_factor(G)
            except np.linalg.LinAlgError:
                broke += 1                                            # a non-positive pivot: chol_factor marks the trial by itself
                continue
            s0 = sl.cho_solve(c, J.T @ y)
            d = sl.cho_solve(c, J.T @ (y - J @ s0))
            piv = np.abs(np.diag(c[0]))
            out.append((G.diagonal().max() * np.abs(s0).max() ** 2 / (y @ y), np.abs(d).max() / np.abs(s0).max(), (piv.max() / piv.min()) ** 2))
        return np.array(out), broke

    print(f'{"run":34s} {"steps":>6s} {"chol broke":>10s} {"growth max / min":>22s} {"refinement max / min":>24s} {"pivot^2 spread max":>19s}')
    for t in range(6):
        q0 = q_goal + rng.uniform(-0.15, 0.15, 7)
        noise = rng.standard_cauchy(size=(299, 32)) if t % 2 else rng.standard_t(3, size=(299, 32)) * 0.5
        x0 = (plant.J * (1 + 0.1 * rng.normal(size=plant.J.shape))).ravel()
        ref = rmckf_block.run_closed_loop(plant.features, q0, des, noise, 0.05, 0.05 * 299.5, 0.2, x0, method='GMCKF', initial_guess=False)
        a, broke = ratios(ref['X'][:-1], ref['err'][1:])
        print(f'{"healthy " + ("Cauchy" if t % 2 else "Student-3") + f" #{t}":34s} {len(a):6d} {broke:10d} {a[:, 0].max():10.3g} / {a[:, 0].min():9.3g} {a[:, 1].max():11.3g} / {a[:, 1].min():10.3g} {a[:, 2].max():19.3g}')
    rng = np.random.default_rng(77)
    qq, _ = np.linalg.qr(rng.normal(size=(32, 7)))
    for c in (1000.0, 300.0, 100.0, 30.0, 10.0):
        Jk = 50.0 * qq @ (np.eye(7) - c * np.triu(np.ones((7, 7)), 1))
        q0 = q_goal + rng.uniform(-0.15, 0.15, 7)
        ref = rmckf_block.run_closed_loop(plant.features, q0, des, rng.standard_t(3, size=(120, 32)) * 0.5, 0.05, 0.05 * 120.5, 0.2, Jk.ravel(), method='GMCKF',
                                          initial_guess=False)
        a, broke = ratios(ref['X'][:-1], ref['err'][1:])
        print(f'{f"Kahan-like c = {c:g}, cond {np.linalg.cond(Jk):.1e}":34s} {len(a):6d} {broke:10d} {a[:, 0].max():10.3g} / {a[:, 0].min():9.3g} {a[:, 1].max():11.3g} / {a[:, 1].min():10.3g} {a[:, 2].max():19.3g}')


if __name__ == '__main__':
    if '--wide' in sys.argv:
        study_wide()
        sys.exit(0)
    print(f'{"fixture":44s} {"max growth":>11s} {"p50 growth":>11s} {"max LS/pinv dev":>16s} {"max cond":>10s} {"max spread":>11s}  steps with dev > 1e-6: their min growth / min spread')
    for name in golden_names('rankdef_') + golden_names('closed_') + golden_names('fpi_'):
        a = study(name)
        if not len(a):
            continue
        bad = a[a[:, 1] > 1e-6]
        print(f'{name:44s} {a[:, 0].max():11.3g} {np.median(a[:, 0]):11.3g} {a[:, 1].max():16.3g} {a[:, 2].max():10.3g} {a[:, 3].max():11.3g}  '
              + (f'{len(bad)} steps: min growth {bad[:, 0].min():.3g}, min spread {bad[:, 3].min():.3g}' if len(bad) else '-'))

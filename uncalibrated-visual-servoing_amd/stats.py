"""Error statistics of a Monte-Carlo sweep, as defined by the reference's MATLAB post-processing
(results/plot_errorbar.m:20-98): per trial ||ISE||_2, ||IAE||_2, ||ITAE||_2 over the features (computed on the GPU,
``engine.closed_loop`` / ``engine.stats_reduce``), FAIL trials dropped (:25), then per sweep cell mean / std / median.
"""
import numpy as np

NAMES = ('ise', 'iae', 'itae')


def cell_summary(stats, status, cell):
    """stats (T, 3), status (T,), cell (T,) integer cell index -> dict per cell of mean/std/median (MATLAB std: N-1)."""
    stats, status, cell = np.asarray(stats, float), np.asarray(status), np.asarray(cell)
    out = {}
    for c in np.unique(cell):
        sel = (cell == c) & (status == 0)
        rows = stats[sel]
        entry = {'trials': int((cell == c).sum()), 'success': int(sel.sum())}
        for j, name in enumerate(NAMES):
            col = rows[:, j]
            entry[name + '_mean'] = float(col.mean()) if len(col) else float('nan')
            entry[name + '_std'] = float(col.std(ddof=1)) if len(col) > 1 else float('nan')
            entry[name + '_median'] = float(np.median(col)) if len(col) else float('nan')
        out[int(c)] = entry
    return out

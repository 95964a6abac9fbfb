"""The reference's behaviour on an infinite measurement, as restated op for op by oracle/rmckf_dense.py: MCKF skips the correction of that step
(Cy = 0, inv raises) and the trial ends one step later; the other estimators lose X at the step itself.  oracle/c must say the same."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle, plant_ref, rmckf_dense


@pytest.mark.parametrize('method,k_fail', [('MCKF', 12), ('GMCKF', 11), ('KF', 11), ('IMCCKF', 11)])
def test_infinite_sample(method, k_fail):
    g = load_golden('closed_mckf_a1p5')
    meta = g['meta']
    noise = g['noise'][:40].copy()
    noise[11, 2] = np.inf
    it = iter(noise)
    with np.errstate(all='ignore'):
        out = rmckf_dense.run_closed_loop(plant_ref.PinholeUR10(meta['dt']), g['q_start'], g['desired'], lambda: next(it), meta['dt'], 0.05 * 40.5, meta['gain'],
                                          method=method, kernel_bw=10, fpi_threshold=0.1, fpi_epoch_max=1000, capture=True)
    ref = c_oracle.closed_loop_batch(g['q_start'][None], noise[None], g['desired'], method, steps=40)
    assert out['status'] == 1 == ref['status'][0] and out['k_done'] == k_fail == ref['k_done'][0]
    if method == 'MCKF':
        assert out['fpi_skipped'][11] and not out['fpi_skipped'][12] and np.all(np.isfinite(out['X'][11])) and ref['fpi'][0, 11] == 0

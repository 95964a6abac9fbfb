"""batch.run_sweep against the reference's EXPERIMENT: the twelve 1 200-trial Monte-Carlo tables the unmodified main.py produced in the build
container (tests/golden/sweep_*.npz, oracle/gen_golden_sweep.py; main.py:104-196 reduced as results/plot_errorbar.m:20-98).  The sweep runs as
the product runs it -- config.json in, device seeding + device noise + closed-loop kernels, per-trial rows out -- and every trial the oracle
reproduces from a 1e-14-moved start must agree with the reference: status and k_done exact, ||ISE|| / ||IAE|| / ||ITAE|| to 1e-8, FAIL counts
per cell, per-cell medians to 1e-6."""
import numpy as np
import pytest

from sweep_common import SWEEPS, check_against_reference, load_sweep

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', SWEEPS)
def test_run_sweep_reproduces_the_reference_experiment(name):
    import uvs_amd as uvs
    ref = load_sweep(name)
    cfg = ref['config']
    res = uvs.batch.run_sweep(cfg)                                   # 12 cells x 100 trials, cell after cell (main.py:121-148)
    assert len(res.pieces) == 12 and res.stats.shape == (1200, 3)
    line, calm = check_against_reference(uvs, name, ref, res.plan, res.stats, res.status, res.k_done, 'batch.run_sweep on the GPU')
    # the same sweep as ONE grid (run_batch: every cell in one launch, per-trial noise generation) returns the same rows bit for bit
    whole = uvs.batch.run_batch(cfg, want=())
    assert np.array_equal(whole.stats.cpu().numpy(), res.stats) and np.array_equal(whole.status.cpu().numpy(), res.status)
    assert np.array_equal(whole.k_done.cpu().numpy(), res.k_done)
    # per-cell table as stats.cell_summary reports it = plot_errorbar.m's, where no trial of the cell is chaotic
    summ = res.cell_summary()
    for c in range(12):
        assert summ[c]['success'] == int((res.status[res.plan.cell == c] == 0).sum())
        if calm[res.plan.cell == c].all():
            assert summ[c]['success'] == int(ref['cell_n_success'][c])
            for j, key in enumerate(('ise', 'iae', 'itae')):
                for what in ('mean', 'std', 'median'):
                    want = float(ref['cell_' + what][c, j])
                    assert abs(summ[c][f'{key}_{what}'] - want) <= 1e-6 * abs(want), (name, c, key, what)

"""Monte-Carlo driver on the GPU: the reference's 12-cell sweep, statistics per cell, results.csv in the reference's format."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def uvs():
    import uvs_amd
    return uvs_amd


def _cfg(method='GMCKF', epoch=20):
    cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_reference.json')))
    cfg['estimator']['method'] = method
    cfg['experiments']['epoch'] = epoch
    return cfg


def test_sweep_of_the_reference_config(uvs):
    """main.py's experiment: 12 alphas x epoch trials.  Heavier tails must cost more ITAE; alpha = 2 sits at the level of the
    paper's figure (results/results1.fig: RMCKF 22 494 +- 2 709 at alpha = 2 on the real simulator)."""
    res = uvs.batch.run_batch(_cfg(epoch=64), want=('err',))
    assert len(res.plan) == 768 and res.stats.shape == (768, 3)
    s = uvs.stats.cell_summary(res.stats.cpu().numpy(), res.status.cpu().numpy(), res.plan.cell)
    itae = np.array([s[c]['itae_median'] for c in range(12)])
    assert all(s[c]['success'] == 64 for c in range(12))
    assert itae[0] > 2 * itae[-1] and np.all(np.diff(itae[[0, 3, 6, 9, 11]]) < 0)
    assert 1.5e4 < s[11]['itae_mean'] < 4e4
    # statistics are those of the error stream
    t = uvs.engine.loop_clock(0.05, 15)
    again = uvs.engine.stats_reduce(res.streams['err'], t, res.k_done).cpu().numpy()
    assert rel_err(again, res.stats.cpu().numpy()) <= 1e-12


def test_first_trial_of_the_sweep_is_the_reference_trial(uvs):
    """Trial 0 of the sweep = alpha 1.0, seed 123456, first jitter draw: the same trial run through Experiment.run()."""
    cfg = _cfg(epoch=2)
    res = uvs.batch.run_batch(cfg, want=('err', 'q'), noise_on_device=False)
    prof = uvs.NoiseProfiler(8, uvs.NoiseType.ALPHA_STABLE, seed=123456, noise_params=dict(cfg['noise']['noise_params'], alpha=1.0))
    ex = uvs.Experiment(res.plan.q_start[0], cfg['experiments']['desired_f'], prof, 0.05, 15, 0.2, uvs.SyntheticRobot(), uvs.Method.GMCKF,
                        method_params=cfg['estimator']['estimator_params'])
    status, t, err, q, *_ = ex.run()
    assert np.array_equal(res.streams['err'].cpu().numpy()[:, :, 0], err) and np.array_equal(res.streams['q'].cpu().numpy()[:, :, 0], q)


def test_results_csv_has_the_reference_format(uvs, tmp_path):
    import pandas as pd
    cfg = _cfg(epoch=1)
    res = uvs.batch.run_batch(cfg, cells=[1.5, 2.0], want=('err', 'q', 'f'))
    path = tmp_path / 'results.csv'
    uvs.batch.write_results_csv(res, cfg, uvs.SyntheticPlant.ur10(cfg['experiments']['desired_f']), str(path))
    df = pd.read_csv(path)
    header = open(os.path.join(ROOT, 'tests', 'golden', 'results_gmckf.csv')).readline().strip().split(',')   # written by the reference's main.py
    assert list(df.columns) == header and len(df) == 2 * 299
    assert set(df['status']) == {'ExperimentStatus.SUCCESS'} and set(df['experiment_id']) == {0, 1} and set(df['kernel_bw']) == {-1.0}
    first = df[df.experiment_id == 0]
    assert np.allclose(first['t'].values, uvs.engine.loop_clock(0.05, 15)) and np.allclose(first['rho'].values, 1.5)
    f = first[[f'f_{i}' for i in range(1, 9)]].values
    d = first[[f'desired_f_{i}' for i in range(1, 9)]].values
    assert np.allclose(f - d, res.streams['err'].cpu().numpy()[:, :, 0], rtol=0, atol=1e-9)
    # the MATLAB post-processing recomputes the statistics from these columns (results/plot_errorbar.m:39-84)
    from oracle.rmckf_dense import trial_stats
    assert rel_err(trial_stats(d - f, first['t'].values), res.stats.cpu().numpy()[0]) <= 1e-9


def test_results_npz_sink(uvs, tmp_path):
    cfg = _cfg(epoch=3)
    res = uvs.batch.run_batch(cfg, cells=[1.0, 1.5], want=('err',))
    path = uvs.batch.save_results_npz(res, cfg, str(tmp_path / 'sweep.npz'))
    z = np.load(path)
    assert list(z['experiment_id']) == list(range(6)) and list(z['rho']) == [1.0] * 3 + [1.5] * 3 and list(z['seed']) == [123456 + i for i in range(6)]
    assert np.array_equal(z['stats'], res.stats.cpu().numpy()) and z['stats'].shape == (6, 3) and list(z['stats_columns']) == ['ise', 'iae', 'itae']
    assert np.array_equal(z['stream_err'], res.streams['err'].cpu().numpy()) and z['stream_err'].shape == (299, 8, 6)
    assert np.array_equal(z['status'], np.zeros(6)) and np.array_equal(z['k_done'], np.full(6, 299)) and len(z['t']) == 299
    import json
    assert json.loads(str(z['config']))['estimator']['method'] == 'GMCKF'


def test_run_batch_launch_options(uvs):
    """batch.run_batch(strict_pinv=..., latency=...): the library's option bits reach the launch; a small sweep stays within the oracle
    gates under either (the options are not config.json keys: the reference's schema is untouched)."""
    import bench
    cfg = bench.config2()
    cfg['experiments']['epoch'] = 300
    base = uvs.batch.run_batch(cfg, cells=[1.5], want=('err',))
    lat = uvs.batch.run_batch(cfg, cells=[1.5], want=('err',), latency=True)
    strict = uvs.batch.run_batch(cfg, cells=[1.5], want=('err',), strict_pinv=True)
    a, b, c = (r.streams['err'].cpu().numpy() for r in (base, lat, strict))
    scale = np.abs(a).max(axis=(0, 1))
    assert not np.array_equal(a, b) and np.median(np.abs(a - b).max(axis=(0, 1)) / scale) <= 1e-12      # four lanes: other last bits
    assert np.median(np.abs(a - c).max(axis=(0, 1)) / scale) <= 1e-12                                   # SVD finish = back substitution to rounding
    for r in (base, lat, strict):
        assert int(r.status.sum()) == 0 and r.stats.shape == (300, 3)
    assert 'strict_pinv' not in cfg and 'latency' not in cfg['experiments'] and 'latency' not in cfg['estimator']


@pytest.mark.parametrize('method,noise_type,hold', [('GMCKF', 'ALPHA_STABLE', False), ('MCKF', 'ALPHA_STABLE', False), ('GMCKF', 'GAUSSIAN_MIXTURE', True)])
def test_run_sweep_is_run_batch_cell_after_cell(uvs, method, noise_type, hold):
    """batch.run_sweep (pieces through one set of buffers, rows copied out on a second stream) returns run_batch's per-trial rows bit for bit:
    ragged pieces (cells of 150 trials cut at 64), shared and per-trial noise generation, a rank's shard that starts inside a cell."""
    cfg = _cfg(method, epoch=150)
    if noise_type != 'ALPHA_STABLE':
        cfg['noise'].update(type=noise_type, hold=hold, hold_time=0.5, noise_params=dict(std=1.0, mean=50.0, rho=0.1))
    cells = [1.0, 1.3, 2.0] if noise_type == 'ALPHA_STABLE' else [0.0, 0.1, 0.2]
    whole = uvs.batch.run_batch(cfg, cells=cells, want=('err',))
    for rank, world in ((0, 1), (1, 4)):
        seen = []
        sw = uvs.batch.run_sweep(cfg, cells=cells, rank=rank, world=world, want=('err',), max_trials=64,
                                 on_piece=lambda a, b, c, out: seen.append((a, b, c, out['err'].clone())))
        lo, hi = sw.lo, sw.hi
        assert (lo, hi) == uvs.dist.shard_range(450, rank, world) and [p[:2] for p in sw.pieces] == [s[:2] for s in seen]
        assert all(b - a <= 64 and len(set(whole.plan.cell[a:b])) == 1 for a, b, _ in sw.pieces) and sum(b - a for a, b, _ in sw.pieces) == hi - lo
        assert np.array_equal(sw.stats, whole.stats.cpu().numpy()[lo:hi]) and np.array_equal(sw.status, whole.status.cpu().numpy()[lo:hi])
        assert np.array_equal(sw.k_done, whole.k_done.cpu().numpy()[lo:hi])
        kd = whole.k_done.cpu().numpy()
        for a, b, c, err in seen:
            ref = whole.streams['err'][:, :, a:b].cpu().numpy()
            got = err.cpu().numpy()
            for j in range(b - a):                                     # rows at and after k_done are unspecified
                assert np.array_equal(got[:kd[a + j], :, j], ref[:kd[a + j], :, j])
        assert sw.rows().shape == (hi - lo, 5) and sw.seconds > 0
    full = uvs.batch.run_sweep(cfg, cells=cells)                       # no streams, whole cells, no callback: the pipelined form
    assert np.array_equal(full.stats, whole.stats.cpu().numpy()) and np.array_equal(full.status, whole.status.cpu().numpy())
    s_a = full.cell_summary()
    s_b = uvs.stats.cell_summary(whole.stats.cpu().numpy(), whole.status.cpu().numpy(), whole.plan.cell)
    assert s_a == s_b or all(np.allclose([s_a[c][k] for k in s_a[c]], [s_b[c][k] for k in s_b[c]], equal_nan=True) for c in s_a)

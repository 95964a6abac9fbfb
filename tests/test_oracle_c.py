"""The plain-C oracle (oracle/c/rmckf_oracle.c) against the reference fixtures and the numpy oracle."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, rel_err
from oracle import c_oracle

CLOSED = golden_names('closed_')
FPI = golden_names('fpi_')
CHAOTIC = {'closed_gmckf_mix_anneal_hold'}


@pytest.mark.parametrize('name', CLOSED)
def test_c_oracle_reproduces_reference(name):
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    out = c_oracle.closed_loop_batch(g['q_start'][None], g['noise'][None], g['desired'], meta['method'], p['kernel_bw'], p['annealing'],
                                     meta['dt'], meta['t_max'], meta['gain'], want_x=True, fpi_threshold=p['fpi_threshold'], fpi_epoch_max=p['fpi_epoch_max'])
    assert out['status'][0] == int(g['status']) and out['k_done'][0] == len(g['t'])
    horizon = 40 if name in CHAOTIC else len(g['t'])
    assert rel_err(out['err'][0, :horizon], g['err'][:horizon]) <= 1e-9
    assert rel_err(out['q'][0, :horizon], g['q'][:horizon]) <= 1e-9
    steps = g['X_steps'][g['X_steps'] < horizon]
    assert rel_err(out['X'][0, steps], g['X'][:len(steps)]) <= 1e-9
    if name not in CHAOTIC:
        from oracle.rmckf_dense import trial_stats
        assert rel_err(out['stats'][0], trial_stats(g['err'], g['t'])) <= 1e-9


def test_c_oracle_fails_on_non_finite_measurement():
    g = load_golden('closed_gmckf_a1p5')
    noise = g['noise'][None, :40].copy()
    noise[0, 11, 2] = np.inf
    out = c_oracle.closed_loop_batch(g['q_start'][None], noise, g['desired'])
    assert out['status'][0] == 1 and out['k_done'][0] == 11


@pytest.mark.parametrize('name', FPI)
def test_c_oracle_reproduces_reference_fpi(name):
    """MCKF with a live fixed-point iteration (oracle/gen_golden_fpi.py): trajectories, passes per step, status and the FAILing step."""
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    out = c_oracle.closed_loop_batch(g['q_start'][None], g['noise_full'][None], g['desired'], 'MCKF', p['kernel_bw'], p['annealing'],
                                     meta['dt'], meta['t_max'], meta['gain'], want_x=True, fpi_threshold=p['fpi_threshold'], fpi_epoch_max=p['fpi_epoch_max'])
    k = len(g['t'])
    assert out['status'][0] == int(g['status']) and out['k_done'][0] == k
    assert np.array_equal(out['fpi'][0, :len(g['fpi_epochs'])], g['fpi_epochs'])
    # cap4: deviations sit at 4e-11 by step 255 and grow 100-fold across an ill-conditioned stretch (cond 577, steps 274-280) -- 6e-9 at the end
    tol = 1e-7 if name == 'fpi_mckf_a1p2_cap4' else 1e-9
    assert rel_err(out['err'][0, :k], g['err']) <= tol and rel_err(out['q'][0, :k], g['q']) <= tol
    assert rel_err(out['X'][0, g['X_steps']], g['X']) <= tol


@pytest.mark.parametrize('name', CLOSED + FPI)
def test_c_oracle_replay_reproduces_reference(name):
    """Open-loop replay in plain C (uvs_oracle_replay): the reference's recorded f / regressor streams in, its per-step X and commands out."""
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    k = len(g['t'])
    f_seq = np.vstack([g['f_init'][None], g['f']])
    out = c_oracle.replay_batch(f_seq[None], g['dq_prev'][None], g['X'][0][None], g['desired'], meta['method'], p['kernel_bw'], p['annealing'],
                                int(meta['t_max'] / meta['dt']), meta['gain'], p['fpi_threshold'], p['fpi_epoch_max'])
    assert out['status'][0] == 0 and out['k_done'][0] == k
    assert rel_err(out['X'][0, g['X_steps']], g['X']) <= 1e-10
    assert rel_err(out['dq_cmd'][0, :-1], g['dq_prev'][1:]) <= 1e-8
    if 'fpi_epochs' in g:
        assert np.array_equal(out['fpi'][0], g['fpi_epochs'][:k])

"""The plain-C oracle (oracle/c/rmckf_oracle.c) against the reference fixtures and the numpy oracle."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, rel_err
from oracle import c_oracle

CLOSED = [n for n in golden_names('closed_') if '_mckf_' not in n]
CHAOTIC = {'closed_gmckf_mix_anneal_hold'}


@pytest.mark.parametrize('name', CLOSED)
def test_c_oracle_reproduces_reference(name):
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    out = c_oracle.closed_loop_batch(g['q_start'][None], g['noise'][None], g['desired'], meta['method'], p['kernel_bw'], p['annealing'],
                                     meta['dt'], meta['t_max'], meta['gain'], want_x=True)
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

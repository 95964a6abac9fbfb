"""GPU parity: the HIP path (through the C ABI) against the reference fixtures and the oracle.  All tests need an MI355X."""
import json

import numpy as np
import pytest

from conftest import golden_names, load_golden, rel_err, scene_desired

pytestmark = pytest.mark.gpu

CLOSED = golden_names('closed_')                                             # KF / MCKF / IMCCKF / GMCKF fixtures
CHAOTIC = {'closed_gmckf_mix_anneal_hold'}                                   # feedback amplifies rounding
LANES_86 = (1, 2, 4, -1, -2, -4, 8, -8)  # 1, 2, 4: tuned kernel (closed loop); 8: the wide kernel's one-row-per-lane DH instantiation (closed loop); negative: generic template with |L| lanes
LANES_CLOSED = (0,) + LANES_86          # 0: the library's own choice (small batches: the four-lane kernels with the two-lane bits)


@pytest.fixture(scope='module')
def uvs():
    import torch
    assert torch.cuda.is_available()
    import uvs_amd
    uvs_amd.lib()
    return uvs_amd


def _cuda(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device='cuda')


def _fp(uvs, g, lanes=0, steps=None, **kw):
    meta, p = g['meta'], g['meta']['params']
    return uvs.engine.make_params(8, 6, meta['method'], p['kernel_bw'], p['annealing'], meta['dt'], meta['t_max'], meta['gain'],
                                  g['desired'], p['initial_guess'], lanes, steps, p['fpi_threshold'], p['fpi_epoch_max'], **kw)


# ---------------------------------------------------------------------------------------------- open-loop replay
@pytest.mark.parametrize('lanes', LANES_86)
@pytest.mark.parametrize('name', CLOSED)
def test_replay_matches_reference(uvs, name, lanes):
    """Feed the reference's recorded f / dq streams through the kernel: per-step X within 1e-10 (contract 1e-5)."""
    g = load_golden(name)
    K = len(g['t'])
    fp = _fp(uvs, g, lanes)
    f_seq = np.vstack([g['f_init'][None], g['f']])                           # (K+1, 8)
    T = 3                                                                    # same trial three times: lanes must agree bitwise
    f = _cuda(np.repeat(f_seq[:, :, None], T, axis=2))
    dq = _cuda(np.repeat(g['dq_prev'][:, :, None], T, axis=2))
    x0 = _cuda(np.tile(g['X'][0], (T, 1)))
    out = uvs.engine.replay(fp, f, dq, x0, final_state=True)
    X = out['x'].cpu().numpy()
    assert np.array_equal(X[:, :, 0], X[:, :, 1]) and np.array_equal(X[:, :, 0], X[:, :, 2])
    assert rel_err(X[g['X_steps'], :, 0], g['X']) <= 1e-10
    assert np.array_equal(out['err'].cpu().numpy()[:, :, 0], g['err'])
    assert rel_err(out['dqcmd'].cpu().numpy()[:-1, :, 0], g['dq_prev'][1:]) <= 1e-9
    if g['meta']['method'] == 'GMCKF':
        kap = out['kappa'].cpu().numpy()[:, :, 0]
        ref = np.exp(-0.5 * g['e'] ** 2 / g['sigma'][:, None] ** 2)
        assert np.all(kap > 0 - 1e-300) and np.all(kap <= 1.0) and rel_err(kap, ref) <= 1e-9
    if int(g['P_steps'][-1]) == K - 1:
        P = out['p_final'].cpu().numpy()[0].reshape(8, 6, 6)
        assert rel_err(P, g['P_blocks'][-1]) <= 1e-10
        assert np.array_equal(P, np.transpose(P, (0, 2, 1)))
    assert int(out['status'].sum()) == 0 and int(out['k_done'][0]) == K


@pytest.mark.parametrize('lanes', [0, 4])
@pytest.mark.parametrize('name', CLOSED)
def test_estimator_only_replay_matches_reference(uvs, name, lanes):
    """No commanded dq requested: the library runs the register-resident estimator kernel (four lanes per filter, blocked lane mapping, two
    wavefronts per SIMD).  Same gates as the full replay, ragged batch (35 trials = 2 full wavefronts of 16 + 3), and agreement with the
    two-lane kernel that also solves the control law."""
    g = load_golden(name)
    K = len(g['t'])
    f_seq = np.vstack([g['f_init'][None], g['f']])
    T = 35
    rng = np.random.default_rng(11)
    f = np.repeat(f_seq[:, :, None], T, axis=2)
    f[1:, :, 1:] += rng.standard_normal((K, 8, T - 1))                        # other trials see other measurements
    dq = np.repeat(g['dq_prev'][:, :, None], T, axis=2)
    x0 = np.tile(g['X'][0], (T, 1))
    out = uvs.engine.replay(_fp(uvs, g, lanes), _cuda(f), _cuda(dq), _cuda(x0), want=('x', 'err', 'kappa'), final_state=True)
    ref = uvs.engine.replay(_fp(uvs, g, 2), _cuda(f), _cuda(dq), _cuda(x0), want=('x', 'err', 'kappa', 'dqcmd'), final_state=True)
    X = out['x'].cpu().numpy()
    assert rel_err(X[g['X_steps'], :, 0], g['X']) <= 1e-10
    assert np.array_equal(out['err'].cpu().numpy()[:, :, 0], g['err'])
    if int(g['P_steps'][-1]) == K - 1:
        P = out['p_final'].cpu().numpy()[0].reshape(8, 6, 6)
        assert rel_err(P, g['P_blocks'][-1]) <= 1e-10 and np.array_equal(P, np.transpose(P, (0, 2, 1)))
    for key in ('x', 'err', 'kappa', 'x_final', 'p_final'):
        assert rel_err(out[key].cpu().numpy(), ref[key].cpu().numpy()) <= 1e-12, key
    assert int(out['status'].sum()) == 0 and np.all(out['k_done'].cpu().numpy() == K)


@pytest.mark.parametrize('method', ['GMCKF', 'KF'])
def test_estimator_only_replay_rowgroup_wavefronts(uvs, method):
    """Library default for KF / RMCKF without the commanded dq: the four row groups of a filter are the four wavefronts of a 64-trial
    workgroup.  Same arithmetic as the four-lane-group kernel, so every stream is bit-identical to it: 150 trials (2 workgroups + 22),
    two trials failing in different row groups at different steps (the earlier step wins)."""
    g = load_golden({'GMCKF': 'closed_gmckf_a1p5', 'KF': 'closed_kf_a1p5'}[method])
    K, T = 80, 150
    rng = np.random.default_rng(12)
    f_seq = np.vstack([g['f_init'][None], g['f']])[:K + 1]
    f = np.repeat(f_seq[:, :, None], T, axis=2)
    f[1:, :, 1:] += rng.standard_normal((K, 8, T - 1))
    f[31, 6, 70] = np.nan                                                      # row group 2 of trial 70 at step 30
    f[51, 1, 70] = np.inf                                                      # row group 1 of the same trial, later
    f[41, 3, 149] = np.inf                                                     # row group 3 of the last (ragged) trial at step 40
    dq = np.repeat(g['dq_prev'][:K, :, None], T, axis=2) * (1.0 + 0.1 * rng.standard_normal((K, 6, T)))
    x0 = np.tile(g['X'][0], (T, 1)) + rng.standard_normal((T, 48))
    want = ('x', 'err', 'kappa')
    a = uvs.engine.replay(_fp(uvs, g, 0, steps=K), _cuda(f), _cuda(dq), _cuda(x0), want=want, final_state=True)
    b = uvs.engine.replay(_fp(uvs, g, 4, steps=K), _cuda(f), _cuda(dq), _cuda(x0), want=want, final_state=True)
    for key in want + ('x_final', 'p_final', 'status', 'k_done'):
        assert np.array_equal(a[key].cpu().numpy(), b[key].cpu().numpy(), equal_nan=True), key
    k_done = a['k_done'].cpu().numpy()
    assert k_done[70] == 30 and k_done[149] == 40 and int((k_done != K).sum()) == 2 and int(a['status'].sum()) == 2


@pytest.mark.parametrize('method', ['GMCKF', 'KF'])
def test_replay_control_wavefronts(uvs, method):
    """Library default for KF / RMCKF with the commanded dq: four estimator wavefronts (row groups) + two control-law wavefronts per 64
    trials (normal equations + refinement).  150 trials (2 workgroups + 22), odd and even horizons, one failing trial: estimator streams
    as the two-lane kernel's, commanded dq within 1e-9 of its Householder solve, status / k_done equal."""
    g = load_golden({'GMCKF': 'closed_gmckf_a1p5', 'KF': 'closed_kf_a1p5'}[method])
    T = 150
    for K in (80, 81, 1):
        rng = np.random.default_rng(14)
        f_seq = np.vstack([g['f_init'][None], g['f']])[:K + 1]
        f = np.repeat(f_seq[:, :, None], T, axis=2)
        f[1:, :, 1:] += rng.standard_normal((K, 8, T - 1))
        if K > 40:
            f[41, 3, 149] = np.inf                                             # the last (ragged) trial fails at step 40
        dq = np.repeat(g['dq_prev'][:K, :, None], T, axis=2) * (1.0 + 0.1 * rng.standard_normal((K, 6, T)))
        x0 = np.tile(g['X'][0], (T, 1)) + rng.standard_normal((T, 48))
        want = ('x', 'err', 'kappa', 'dqcmd')
        a = uvs.engine.replay(_fp(uvs, g, 0, steps=K), _cuda(f), _cuda(dq), _cuda(x0), want=want, final_state=True)
        b = uvs.engine.replay(_fp(uvs, g, 2, steps=K), _cuda(f), _cuda(dq), _cuda(x0), want=want, final_state=True)
        good = np.arange(T) != 149
        for key in ('x', 'err', 'kappa'):
            assert rel_err(a[key].cpu().numpy()[:, :, good], b[key].cpu().numpy()[:, :, good]) <= 1e-12, (key, K)
        assert rel_err(a['dqcmd'].cpu().numpy()[:, :, good], b['dqcmd'].cpu().numpy()[:, :, good]) <= 1e-9, K
        for key in ('x_final', 'p_final'):
            assert rel_err(a[key].cpu().numpy()[good], b[key].cpu().numpy()[good]) <= 1e-12, (key, K)
        assert np.array_equal(a['status'].cpu().numpy(), b['status'].cpu().numpy()) and np.array_equal(a['k_done'].cpu().numpy(), b['k_done'].cpu().numpy())
        if K > 40:
            assert int(a['k_done'][149]) == 40 and int(a['status'][149]) == 1 and int(a['status'].sum()) == 1


def test_replay_default_kernels_full_horizon(uvs):
    """4 096 trials x 299 steps through the three default replay mappings (row-group wavefronts, + control wavefronts, record path) and the
    two-lane kernel: estimator streams equal to 1e-12, commanded dq to 1e-9 (normal equations vs Householder), no trial fails."""
    import torch
    g = load_golden('closed_gmckf_a1p5')
    T, K = 4096, 299
    gen = torch.Generator(device='cuda').manual_seed(21)
    rnd = lambda *shape: torch.randn(shape, device='cuda', dtype=torch.float64, generator=gen)      # noqa: E731
    J = rnd(8, 6, T) * 50
    dq = rnd(K, 6, T) * 0.2
    f = torch.empty((K + 1, 8, T), device='cuda', dtype=torch.float64)
    f[0] = 128 + 20 * rnd(8, T)
    for k in range(K):
        f[k + 1] = f[k] + torch.einsum('mnt,nt->mt', J, dq[k]) * 0.05 + rnd(8, T)
    x0 = (J + 5 * rnd(8, 6, T)).permute(2, 0, 1).reshape(T, 48).contiguous()
    ref = uvs.engine.replay(_fp(uvs, g, 2, steps=K), f, dq, x0, want=('x', 'err', 'dqcmd'), final_state=True)
    a = uvs.engine.replay(_fp(uvs, g, 0, steps=K), f, dq, x0, want=('x', 'err', 'dqcmd'), final_state=True)
    b = uvs.engine.replay(_fp(uvs, g, 0, steps=K), f, dq, x0, want=('x', 'err'), final_state=True)
    c = uvs.engine.replay(_fp(uvs, g, 0, steps=K), f, dq, x0, want=('x', 'err'), layout='ktc', in_layout='kct', final_state=True)
    for out, lay in ((a, 'kct'), (b, 'kct'), (c, 'ktc')):
        for key in ('x', 'err'):
            d = (uvs.engine.as_tkc(out[key], lay) - uvs.engine.as_tkc(ref[key], 'kct')).abs().max() / uvs.engine.as_tkc(ref[key], 'kct').abs().max()
            assert float(d) <= 1e-12, (key, lay, float(d))
        assert float((out['p_final'] - ref['p_final']).abs().max() / ref['p_final'].abs().max()) <= 1e-12
        assert int(out['status'].sum()) == 0 and bool((out['k_done'] == K).all())
    d = (a['dqcmd'] - ref['dqcmd']).abs().max() / ref['dqcmd'].abs().max()
    assert float(d) <= 1e-9, float(d)


def test_replay_default_kernels_edge_sizes(uvs):
    """The library-default replay mappings (row-group wavefronts, + control wavefronts with the commanded dq, record path) at the edges:
    one trial, one short of / exactly / one past a 64-trial workgroup, horizons of 1, 2 and 3 steps -- against the two-lane kernel."""
    g = load_golden('closed_gmckf_a1p5')
    rng = np.random.default_rng(15)
    for T in (1, 16, 63, 64, 65, 129):
        for K in (1, 2, 3, 17):
            f_seq = np.vstack([g['f_init'][None], g['f']])[:K + 1]
            f = np.repeat(f_seq[:, :, None], T, axis=2) + rng.standard_normal((K + 1, 8, T))
            dq = np.repeat(g['dq_prev'][:K, :, None], T, axis=2) * (1.0 + 0.1 * rng.standard_normal((K, 6, T)))
            x0 = np.tile(g['X'][0], (T, 1)) + rng.standard_normal((T, 48))
            ref = uvs.engine.replay(_fp(uvs, g, 2, steps=K), _cuda(f), _cuda(dq), _cuda(x0), want=('x', 'err', 'dqcmd'), final_state=True)
            a = uvs.engine.replay(_fp(uvs, g, 0, steps=K), _cuda(f), _cuda(dq), _cuda(x0), want=('x', 'err', 'dqcmd'), final_state=True)
            b = uvs.engine.replay(_fp(uvs, g, 0, steps=K), _cuda(f), _cuda(dq), _cuda(x0), want=('x', 'err'), final_state=True)
            c = uvs.engine.replay(_fp(uvs, g, 0, steps=K), _cuda(f), _cuda(dq), _cuda(x0), want=('x', 'err'), layout='ktc', in_layout='kct', final_state=True)
            for out, lay in ((a, 'kct'), (b, 'kct'), (c, 'ktc')):
                for key in ('x', 'err'):
                    assert rel_err(uvs.engine.as_tkc(out[key], lay).cpu().numpy(), uvs.engine.as_tkc(ref[key], 'kct').cpu().numpy()) <= 1e-12, (T, K, key, lay)
                for key in ('x_final', 'p_final'):
                    assert rel_err(out[key].cpu().numpy(), ref[key].cpu().numpy()) <= 1e-12, (T, K, key)
                assert int(out['status'].sum()) == 0 and np.all(out['k_done'].cpu().numpy() == K), (T, K)
            assert rel_err(a['dqcmd'].cpu().numpy(), ref['dqcmd'].cpu().numpy()) <= 1e-9, (T, K)


@pytest.mark.parametrize('T', [48, 35])
@pytest.mark.parametrize('method', ['GMCKF', 'KF', 'IMCCKF', 'MCKF'])
def test_estimator_only_replay_record_layout(uvs, method, T):
    """X and err as per-trial records ([step][trial][component]): whole wavefronts (T = 48) take the LDS-transposed 1 KB-store path, a ragged
    batch (T = 35) the strided one; inputs trial-fastest or records.  Same arithmetic: every stream bit-identical to the trial-fastest run."""
    g = load_golden({'GMCKF': 'closed_gmckf_a1p5', 'KF': 'closed_kf_a1p5', 'IMCCKF': 'closed_imcckf_a1p5', 'MCKF': 'closed_mckf_a1p5'}[method])
    K = 60
    rng = np.random.default_rng(13)
    f_seq = np.vstack([g['f_init'][None], g['f']])[:K + 1]
    f = np.repeat(f_seq[:, :, None], T, axis=2)
    f[1:, :, 1:] += rng.standard_normal((K, 8, T - 1))
    f[21, 2, T - 2] = np.inf                                                   # one trial fails at step 20
    dq = np.repeat(g['dq_prev'][:K, :, None], T, axis=2) * (1.0 + 0.1 * rng.standard_normal((K, 6, T)))
    x0 = np.tile(g['X'][0], (T, 1)) + rng.standard_normal((T, 48))
    fp = _fp(uvs, g, 4, steps=K)
    ref = uvs.engine.replay(fp, _cuda(f), _cuda(dq), _cuda(x0), want=('x', 'err'), final_state=True)
    for in_layout in ('kct', 'ktc'):
        fi, di = (f, dq) if in_layout == 'kct' else (f.transpose(0, 2, 1), dq.transpose(0, 2, 1))
        out = uvs.engine.replay(fp, _cuda(np.ascontiguousarray(fi)), _cuda(np.ascontiguousarray(di)), _cuda(x0), want=('x', 'err'), layout='ktc',
                                final_state=True, in_layout=in_layout)
        for key in ('x', 'err'):
            a = uvs.engine.as_tkc(out[key], 'ktc').cpu().numpy()
            b = uvs.engine.as_tkc(ref[key], 'kct').cpu().numpy()
            assert np.array_equal(a, b, equal_nan=True), (key, in_layout)
        for key in ('x_final', 'p_final', 'status', 'k_done'):
            assert np.array_equal(out[key].cpu().numpy(), ref[key].cpu().numpy(), equal_nan=True), (key, in_layout)
    if method == 'MCKF':                     # Cy = exp(-inf) = 0: inv(Cy) raises, the two steps that see the infinite feature keep the prediction
        assert int(ref['k_done'][T - 2]) == K and int(ref['status'].sum()) == 0 and bool(torch_isfinite_all(ref['x_final']))
    else:
        assert int(ref['k_done'][T - 2]) == 20 and int((ref['status'] == 1).sum()) == 1


def torch_isfinite_all(t):
    import torch
    return torch.isfinite(t).all()


def test_estimator_only_replay_fail_semantics(uvs):
    g = load_golden('closed_gmckf_a1p5')
    K, T, bad = 60, 20, 23
    f_seq = np.vstack([g['f_init'][None], g['f']])[:K + 1]
    f = np.repeat(f_seq[:, :, None], T, axis=2)
    f[bad + 1, 5, 17] = np.inf                                               # trial 17 sees a non-finite feature at step `bad`
    out = uvs.engine.replay(_fp(uvs, g, 0, steps=K), _cuda(f), _cuda(np.repeat(g['dq_prev'][:K, :, None], T, axis=2)),
                            _cuda(np.tile(g['X'][0], (T, 1))), want=('x', 'err'))
    status, k_done = out['status'].cpu().numpy(), out['k_done'].cpu().numpy()
    assert status[17] == 1 and k_done[17] == bad and status.sum() == 1 and np.all(np.delete(k_done, 17) == K)
    X = out['x'].cpu().numpy()
    assert np.array_equal(X[:, :, 0], X[:, :, 19]) and np.array_equal(X[:bad, :, 17], X[:bad, :, 0])


# ---------------------------------------------------------------------------------------------- closed loop
@pytest.mark.parametrize('lanes', LANES_CLOSED)
@pytest.mark.parametrize('name', CLOSED)
def test_closed_loop_matches_reference(uvs, name, lanes):
    """Whole trial in the kernel (plant + estimator + control) on the reference's noise: trajectories within 1e-8."""
    g = load_golden(name)
    K = len(g['t'])
    fp = _fp(uvs, g, lanes)
    plant = uvs.SyntheticPlant.ur10(scene_desired(g))
    out = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(g['q_start'][None]), _cuda(g['noise'][:, :, None]),
                                 want=('x', 'err', 'q', 'f', 'dq'))
    assert int(out['status'][0]) == int(g['status']) and int(out['k_done'][0]) == K
    horizon = 40 if name in CHAOTIC else K
    tol = 1e-8
    err, q, X = (out[k].cpu().numpy()[:, :, 0] for k in ('err', 'q', 'x'))
    assert rel_err(err[:horizon], g['err'][:horizon]) <= tol
    assert rel_err(q[:horizon], g['q'][:horizon]) <= tol
    assert rel_err(out['f'].cpu().numpy()[:horizon, :, 0], g['f'][:horizon]) <= tol
    steps = g['X_steps'][g['X_steps'] < horizon]
    assert rel_err(X[steps], g['X'][:len(steps)]) <= tol
    assert rel_err(out['dq'].cpu().numpy()[:horizon - 1, :, 0], g['dq_prev'][1:horizon]) <= 1e-7
    if name not in CHAOTIC:
        from oracle.rmckf_dense import trial_stats
        assert rel_err(out['stats'].cpu().numpy()[0], trial_stats(g['err'], g['t'])) <= 1e-8


def test_initial_guess_matches_reference(uvs):
    g = load_golden('closed_gmckf_a1p5_jitter')
    fp = _fp(uvs, g, steps=1)
    plant = uvs.SyntheticPlant.ur10(scene_desired(g))
    out = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(g['q_start'][None]), _cuda(g['noise'][:1, :, None]), want=('x', 'f'))
    assert rel_err(out['x'].cpu().numpy()[0, :, 0], g['X'][0]) <= 1e-12      # step 0: H = 0 so X stays X0
    assert rel_err(out['f'].cpu().numpy()[0, :, 0], g['f'][0]) <= 1e-13


# ---------------------------------------------------------------------------------------------- drop-in API
@pytest.mark.parametrize('name', ['closed_gmckf_a1p5', 'closed_gmckf_mix_anneal', 'closed_kf_a2p0', 'closed_imcckf_a1p5', 'closed_mckf_a1p5'])
def test_experiment_api_drop_in(uvs, name):
    """Experiment(...).run() with the reference's call signature returns the reference's 9-tuple."""
    g = load_golden(name)
    meta = g['meta']
    NT, M = uvs.NoiseType, uvs.Method
    prof = uvs.NoiseProfiler(num_features=8, noise_type=NT[meta['noise_type']], seed=meta['seed'], noise_hold=meta['hold'],
                             noise_hold_cnt=meta['hold_cnt'], noise_params=meta['noise_params'])
    ex = uvs.Experiment(q_start=g['q_start'], desired_f=g['desired'], noise_prof=prof, t_s=meta['dt'], t_max=meta['t_max'],
                        ibvs_gain=meta['gain'], robot=uvs.SyntheticRobot(dt=meta['dt']), method=M[meta['method']],
                        method_params=meta['params'])
    status, t, err, q, f, fd, cam, noise, bw = ex.run()
    assert status == uvs.ExperimentStatus.SUCCESS and status.value == int(g['status'])
    assert np.array_equal(t, g['t']) and np.array_equal(noise, g['noise'])
    for got, ref in ((err, g['err']), (q, g['q']), (f, g['f']), (cam[:, :3], g['cam'][:, :3])):
        assert got.shape == ref.shape and rel_err(got, ref) <= 1e-8
    # the generator's plant logged zero angles; computePose's roll / pitch / yaw (ur10_simulation.py:151-163) are checked against the
    # oracle's kinematics in the reference's quat2euler convention
    from oracle import plant_ref
    for k in (0, 100, len(t) - 1):
        R = plant_ref.fkine_all(g['q'][k])[5][:3, :3]
        ref = [np.arctan2(R[2, 1], R[2, 2]), np.arcsin(-R[2, 0]), np.arctan2(R[1, 0], R[0, 0])]
        assert cam.shape == (len(t), 6) and np.allclose(cam[k, 3:], ref, rtol=0, atol=1e-8)
    assert np.array_equal(fd, np.tile(g['desired'], (len(t), 1))) and np.array_equal(bw, g['sigma_log'])


def test_experiment_api_with_external_robot(uvs):
    """A robot the package knows nothing about (the oracle's duck-typed plant): per-step estimator on the GPU."""
    from oracle.plant_ref import PinholeUR10
    g = load_golden('closed_gmckf_a1p5_anneal')
    meta = g['meta']
    prof = uvs.NoiseProfiler(8, uvs.NoiseType.ALPHA_STABLE, seed=meta['seed'], noise_params=meta['noise_params'])
    ex = uvs.Experiment(g['q_start'], g['desired'], prof, meta['dt'], meta['t_max'], meta['gain'], PinholeUR10(meta['dt']),
                        uvs.Method.GMCKF, **meta['params'])
    status, t, err, q, f, fd, cam, noise, bw = ex.run()
    assert status == uvs.ExperimentStatus.SUCCESS and len(t) == 299
    assert np.array_equal(t, g['t']) and np.array_equal(noise, g['noise'])
    assert rel_err(err, g['err']) <= 1e-8 and rel_err(q, g['q']) <= 1e-8 and rel_err(cam, g['cam']) <= 1e-8


def test_analytical_is_refused_loudly(uvs):
    with pytest.raises(NotImplementedError):
        uvs.Experiment([0] * 6, [0] * 8, None, 0.05, 15, 0.2, uvs.SyntheticRobot(), uvs.Method.ANALYTICAL).run()


@pytest.mark.parametrize('lanes', [0, 2, -1, -2, 4, 8])
def test_mckf_fixed_point_iterations_match_block_oracle(uvs, lanes):
    """A tight threshold forces several fixed-point passes (Cholesky factor, Cx != I) and an epoch cap that skips corrections.  Lanes 0 / 2:
    the tuned kernels run the first pass only, mark the trial, and the library's second pass (generic template) iterates."""
    from oracle import rmckf_block
    g = load_golden('closed_mckf_a1p5')
    meta = g['meta']
    K = 120
    f_seq = np.vstack([g['f_init'][None], g['f']])[:K + 1]
    for thr, cap in ((1e-9, 1000), (1e-3, 2)):
        fp = uvs.engine.make_params(8, 6, 'MCKF', 5.0, True, meta['dt'], meta['t_max'], meta['gain'], g['desired'], True, lanes, K, thr, cap)
        out = uvs.engine.replay(fp, _cuda(f_seq[:, :, None]), _cuda(g['dq_prev'][:K, :, None]), _cuda(g['X'][0][None]), final_state=True)
        ref = rmckf_block.run_replay(f_seq, g['dq_prev'][:K], g['X'][0], g['desired'], meta['gain'], 'MCKF', 5.0, True, 300, thr, cap)
        assert ref['fpi_iterations'].max() >= 2
        assert rel_err(out['x'].cpu().numpy()[:, :, 0], ref['X']) <= 1e-9
        assert rel_err(out['p_final'].cpu().numpy()[0].reshape(8, 6, 6), ref['P_final']) <= 1e-9


# ---------------------------------------------------------------------------------------------- the reference's tests/*.py
@pytest.mark.parametrize('lanes', [0, -1])
@pytest.mark.parametrize('name', golden_names('script_'))
def test_replay_matches_reference_scripts(uvs, name, lanes):
    """(2,6) and (6,6) filters of tests/kalman_{1,3}_feature(s).py and tests/mckf_{1,3}_feature(s).py (BASELINE.json configs[0]):
    the streams those scripts produced go through the replay kernel; X per step, final P and the twist command must come back."""
    g = load_golden(name)
    meta = g['meta']
    m, K, mask = meta['m'], meta['steps'], g['cmd_mask']
    fp = uvs.engine.make_params(m, 6, meta['method'], meta['kernel_bw'] or 10.0, False, meta['dt'], 100.0, meta['gain'], g['desired'], False,
                                lanes, K, meta['fpi_threshold'] or 0.1, max(meta['fpi_epoch_max'], 1))
    T = 2
    rep = lambda a: _cuda(np.repeat(a[:, :, None], T, axis=2))               # noqa: E731
    out = uvs.engine.replay(fp, rep(g['f']), rep(g['dp_prev']), _cuda(np.tile(g['X0'], (T, 1))), final_state=True)
    X = out['x'].cpu().numpy()
    assert np.array_equal(X[:, :, 0], X[:, :, 1])
    assert rel_err(X[g['X_steps'], :, 0], g['X']) <= 1e-10
    cmd = out['dqcmd'].cpu().numpy()[:-1, :, 0]
    assert rel_err(cmd[:, mask], g['dp_prev'][1:][:, mask]) <= 1e-8
    assert rel_err(out['p_final'].cpu().numpy()[0].reshape(m, 6, 6), g['P_blocks'][-1]) <= 1e-9
    assert np.array_equal(out['err'].cpu().numpy()[:, :, 0], g['f'][1:] - g['desired'])
    assert int(out['status'].sum()) == 0 and int(out['k_done'][0]) == K


def test_baseline_config_1_batch(uvs):
    """BASELINE.json configs[0] as a closed loop: one feature (m = 2), KF, WHITE_NOISE std 1, 100 trials
    with seeds 123456 + t, analytic initial guess; kernel (2,6) against the per-row oracle on the oracle's own plant."""
    from oracle import noise_ref, plant_ref, rmckf_block, rmckf_dense
    T, dt, t_max, gain = 100, 0.05, 15, 0.2
    desired = np.array([128.0, 128.0])
    plant = uvs.SyntheticPlant.ur10(desired)
    discs = plant_ref.place_discs(desired)
    assert np.allclose(discs, plant.points, rtol=0, atol=1e-15)
    rng = np.random.default_rng(7)
    q0 = plant_ref.Q_GOAL + 0.05 * rng.standard_normal((T, 6))
    K = len(uvs.engine.loop_clock(dt, t_max))
    noise = uvs.noise_batch(uvs.NoiseType.WHITE_NOISE, {'std': 1.0}, 123456 + np.arange(T), 2, K)        # (T, K, 2)
    ref_stream = noise_ref.NoiseStreamRef(2, noise_ref.WHITE_NOISE, 123456 + 3, std=1.0)
    assert np.array_equal(noise[3], ref_stream.take(K))
    fp = uvs.engine.make_params(2, 6, 'KF', 10.0, False, dt, t_max, gain, desired, True)
    out = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0), _cuda(noise.transpose(1, 2, 0)), want=('x', 'err', 'q'))
    assert int(out['status'].sum()) == 0 and int(out['k_done'].min()) == K
    worst = 0.0
    for t in range(0, T, 9):
        robot = plant_ref.PinholeUR10(dt, discs)
        robot.start(q0[t])
        x0 = rmckf_dense.analytic_initial_guess(robot, robot.features(), 2, 6)
        ref = rmckf_block.run_closed_loop(lambda q: plant_ref.project(plant_ref.fkine_all(q)[5], discs), q0[t], desired, noise[t], dt, t_max,
                                          gain, x0, method='KF')
        assert ref['status'] == 0
        worst = max(worst, rel_err(out['err'].cpu().numpy()[:, :, t], ref['err']), rel_err(out['q'].cpu().numpy()[:, :, t], ref['q']),
                    rel_err(out['x'].cpu().numpy()[:, :, t], ref['X']))
        assert rel_err(out['stats'].cpu().numpy()[t], rmckf_dense.trial_stats(ref['err'], ref['t'])) <= 1e-8
    assert worst <= 1e-8


# ---------------------------------------------------------------------------------------------- other shapes
def _random_replay_case(m, n, K, T, seed):
    rng = np.random.default_rng(seed)
    J = rng.normal(size=(T, m, n)) * 20
    dq = rng.normal(size=(T, K, n)) * 0.3
    f = np.zeros((T, K + 1, m))
    f[:, 0] = rng.uniform(60, 200, (T, m))
    for k in range(K):
        f[:, k + 1] = f[:, k] + np.einsum('tmn,tn->tm', J, dq[:, k]) * 0.05 + rng.standard_t(2, size=(T, m))
    x0 = (J + rng.normal(size=J.shape)).reshape(T, m * n)
    return f, dq, x0, rng.uniform(80, 180, m)


@pytest.mark.parametrize('method', ['GMCKF', 'KF', 'IMCCKF'])
@pytest.mark.parametrize('m,n,lanes', [(2, 6, 1), (6, 6, 1), (8, 6, 1), (8, 6, 4), (32, 7, 8), (32, 7, 16), (32, 7, 32)])
def test_replay_other_shapes_match_block_oracle(uvs, m, n, lanes, method):
    """Shapes the reference cannot run (experiment.py hard-wires m = 8, n = 6): oracle = per-row restatement."""
    from oracle import rmckf_block
    K, T = 40, 5
    f, dq, x0, des = _random_replay_case(m, n, K, T, 1000 + m)
    fp = uvs.engine.make_params(m, n, method, 7.5, True, 0.05, 15, 0.2, des, False, lanes, steps=K)
    out = uvs.engine.replay(fp, _cuda(f.transpose(1, 2, 0)), _cuda(dq.transpose(1, 2, 0)), _cuda(x0), final_state=True)
    for t in range(T):
        ref = rmckf_block.run_replay(f[t], dq[t], x0[t], des, 0.2, method, 7.5, True, 300)
        assert rel_err(out['x'].cpu().numpy()[:, :, t], ref['X']) <= 1e-10
        assert rel_err(out['dqcmd'].cpu().numpy()[:, :, t], ref['dq_cmd']) <= 1e-8
        assert rel_err(out['kappa'].cpu().numpy()[:, :, t], ref['kappa']) <= 1e-9
        assert rel_err(out['p_final'].cpu().numpy()[t].reshape(m, n, n), ref['P_final']) <= 1e-10


@pytest.mark.parametrize('method', ['GMCKF', 'KF', 'IMCCKF'])
@pytest.mark.parametrize('lanes', [0, 8, 16, 32, -8, -16])
def test_closed_loop_stress_plant(uvs, lanes, method):
    """(m, n) = (32, 7) on the linear consistent plant (BASELINE config 5), closed loop, against the block oracle.  lanes 0 / 8 / 16: the
    tuned wide-shape kernel (normal-equation control law); 32 and negative values: the generic template (Householder QR)."""
    from oracle import rmckf_block
    plant = uvs.LinearPlant.random(32, 7, seed=2)
    K, T = (299, 3) if lanes in (0, 8, -16) else (80, 4)                      # whole 299-step horizon on the shipped kernel and one generic variant
    rng = np.random.default_rng(5)
    q_goal = plant.q0 + rng.uniform(-0.3, 0.3, 7)
    des = plant.features(q_goal)
    q0 = q_goal + rng.uniform(-0.15, 0.15, (T, 7))
    noise = rng.standard_t(3, size=(T, K, 32)) * 0.5
    x0 = np.tile((plant.J * (1 + 0.1 * rng.normal(size=plant.J.shape))).ravel(), (T, 1))
    fp = uvs.engine.make_params(32, 7, method, 10.0, False, 0.05, 15, 0.2, des, False, lanes, steps=K)
    out = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0), _cuda(noise.transpose(1, 2, 0)), _cuda(x0), want=('x', 'err', 'q'))
    assert not out['status'].cpu().numpy().any() and np.all(out['k_done'].cpu().numpy() == K)
    for t in range(T):
        ref = rmckf_block.run_closed_loop(plant.features, q0[t], des, noise[t], 0.05, 0.05 * (K + 0.5), 0.2, x0[t], method=method, initial_guess=False)
        assert ref['k_done'] == K and ref['status'] == 0
        assert rel_err(out['err'].cpu().numpy()[:, :, t], ref['err']) <= 1e-8
        assert rel_err(out['q'].cpu().numpy()[:, :, t], ref['q']) <= 1e-8
        assert rel_err(out['x'].cpu().numpy()[:, :, t], ref['X']) <= 1e-8


def test_closed_loop_stress_plant_record_layout(uvs):
    """Per-trial records ([step][trial][component]): the wide-shape kernel sends X through its LDS transposition (1 KB stores).  Same
    numbers as the trial-fastest layout bit for bit, whole wavefronts (T = 24) and a ragged batch (T = 21, plain store path) alike."""
    plant = uvs.LinearPlant.random(32, 7, seed=2)
    K = 40
    rng = np.random.default_rng(8)
    q_goal = plant.q0 + rng.uniform(-0.3, 0.3, 7)
    des = plant.features(q_goal)
    for T in (24, 21):
        q0 = q_goal + rng.uniform(-0.15, 0.15, (T, 7))
        noise = rng.standard_t(3, size=(K, 32, T)) * 0.5
        x0 = np.tile((plant.J * (1 + 0.1 * rng.normal(size=plant.J.shape))).ravel(), (T, 1))
        fp = uvs.engine.make_params(32, 7, 'GMCKF', 10.0, True, 0.05, 15, 0.2, des, False, 0, steps=K)
        a = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0), _cuda(noise), _cuda(x0), want=('x', 'err', 'q'), layout='kct')
        b = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0), _cuda(noise.transpose(0, 2, 1)), _cuda(x0), want=('x', 'err', 'q'), layout='ktc')
        for key in ('x', 'err', 'q'):
            assert np.array_equal(uvs.engine.as_tkc(a[key], 'kct').cpu().numpy(), uvs.engine.as_tkc(b[key], 'ktc').cpu().numpy()), (T, key)
        assert np.array_equal(a['stats'].cpu().numpy(), b['stats'].cpu().numpy()) and not b['status'].cpu().numpy().any()


@pytest.mark.parametrize('method', ['KF', 'IMCCKF'])
def test_pair_stores_write_the_same_x_stream(uvs, method):
    """Round 6: the (8,6) two-lane KF / IMCC-KF kernels write a trial-fastest X stream as 16-byte pairs of consecutive trials out of LDS (even T and row
    pitch; profiles/r06/pair_stores_ab.txt).  Same values, other store instructions: the stream equals, bit for bit, the one written through a [trial][step]
    [component] view (plain 8-byte stores) -- whole wavefronts, a ragged last wavefront (T = 70), a batch of one pair, and beyond one round of wavefronts."""
    import bench
    cfg = bench.config2()
    des = cfg['experiments']['desired_f']
    plant = uvs.SyntheticPlant.ur10(des).to_struct()
    for T, K in ((2, 40), (70, 299), (4098, 120), (33000, 30)):
        cfg['experiments']['epoch'] = T
        plan = uvs.batch.plan_trials(cfg, cells=[1.5])
        noise = uvs.batch.device_noise(cfg, plan, 0, T, K, 'cuda', share=False)
        q0 = _cuda(plan.q_start)
        fp = uvs.engine.make_params(8, 6, method, 10, False, 0.05, 15, 0.2, des, True, 2, K)
        a = uvs.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))                      # x: [step][component][trial] -> pair stores
        b = uvs.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'), x_layout='tkc')      # x: [trial][step][component] -> 8-byte stores
        assert np.array_equal(uvs.engine.as_tkc(a['x'], 'kct').cpu().numpy(), uvs.engine.as_tkc(b['x'], 'tkc').cpu().numpy()), (method, T)
        for key in ('err', 'q', 'stats', 'status', 'k_done'):
            assert np.array_equal(a[key].cpu().numpy(), b[key].cpu().numpy()), (method, T, key)


# ---------------------------------------------------------------------------------------------- failure semantics
def test_non_finite_state_fails_the_trial_only(uvs):
    g = load_golden('closed_gmckf_a1p5')
    K, T, bad_step = 50, 6, 17
    fp = _fp(uvs, g, steps=K)
    plant = uvs.SyntheticPlant.ur10(scene_desired(g))
    noise = np.repeat(g['noise'][:K, :, None], T, axis=2)
    noise[bad_step, 3, 2] = np.nan                                           # trial 2 sees a NaN measurement
    out = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(np.tile(g['q_start'], (T, 1))), _cuda(noise), want=('err', 'x'))
    status, k_done = out['status'].cpu().numpy(), out['k_done'].cpu().numpy()
    assert list(status) == [0, 0, 1, 0, 0, 0] and k_done[2] == bad_step and np.all(np.delete(k_done, 2) == K)
    err = out['err'].cpu().numpy()
    assert np.array_equal(err[:, :, 0], err[:, :, 5]) and rel_err(err[:, :, 0], g['err'][:K]) <= 1e-8
    assert np.array_equal(err[:bad_step, :, 2], err[:bad_step, :, 0])      # rows >= k_done are unspecified (the reference trims them)


def test_empty_and_degenerate_batches(uvs):
    """T = 0 is an argument error (nothing to launch); K = 0 runs no step: SUCCESS, zero rows, zero statistics; a single trial
    and a ragged batch (not a multiple of the 32 trials of a wavefront) behave like slices of a bigger batch."""
    import ctypes as C
    g = load_golden('closed_gmckf_a1p5')
    plant = uvs.SyntheticPlant.ur10(scene_desired(g))
    fp = _fp(uvs, g, steps=0)
    out = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(np.tile(g['q_start'], (5, 1))), None, want=('err',))
    assert out['k_done'].tolist() == [0] * 5 and out['status'].tolist() == [0] * 5 and float(out['stats'].abs().max()) == 0.0
    V = uvs._lib.NULL_VIEW
    rc = uvs.lib().uvs_rmckf_closed_loop_f64(C.byref(fp), C.byref(plant.to_struct()), 0, V, V, V, V, V, V, V, V, None, None, None, V, V, None)
    assert rc == -1
    K = 30
    fp = _fp(uvs, g, steps=K)
    q0 = np.tile(g['q_start'], (37, 1))
    q0[:, 1] -= np.linspace(0, 0.3, 37)
    noise = np.random.default_rng(4).standard_t(3, size=(K, 8, 37))
    big = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0), _cuda(noise), want=('err', 'x'))
    one = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0[36:37]), _cuda(noise[:, :, 36:37]), want=('err', 'x'))
    assert np.array_equal(big['err'].cpu().numpy()[:, :, 36], one['err'].cpu().numpy()[:, :, 0])
    assert np.array_equal(big['x'].cpu().numpy()[:, :, 36], one['x'].cpu().numpy()[:, :, 0])
    assert np.array_equal(big['stats'].cpu().numpy()[36], one['stats'].cpu().numpy()[0])


# ---------------------------------------------------------------------------------------------- statistics kernel
def test_stats_kernel_matches_matlab_definition(uvs):
    from oracle.rmckf_dense import trial_stats
    rng = np.random.default_rng(3)
    K, T = 299, 37
    err = rng.normal(size=(T, K, 8)) * 10
    t = uvs.engine.loop_clock(0.05, 15)
    k_done = rng.integers(1, K + 1, T).astype(np.int32)
    got = uvs.engine.stats_reduce(_cuda(err.transpose(1, 2, 0)), t, _cuda(k_done)).cpu().numpy()
    for j in range(T):
        assert rel_err(got[j], trial_stats(err[j, :k_done[j]], t[:k_done[j]])) <= 1e-12


# ---------------------------------------------------------------------------------------------- bulk check against the C oracle
def test_monte_carlo_batch_matches_c_oracle(uvs):
    """2 048 trials of BASELINE config 2 (alpha-stable alpha = 1.5, jittered starts, product noise generator) through the
    default kernel and through oracle/c: trajectories, statistics, status."""
    from oracle import c_oracle
    import bench
    cfg = bench.config2()
    cfg['experiments']['epoch'] = 2048
    plan = uvs.batch.plan_trials(cfg, cells=[1.5])
    K = 299
    noise = np.zeros((len(plan), K, 8))
    uvs.batch.trial_noise(cfg, plan, 0, len(plan), K, noise)
    fp = uvs.engine.make_params(8, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True)
    out = uvs.engine.closed_loop(fp, uvs.SyntheticPlant.ur10().to_struct(), _cuda(plan.q_start), _cuda(noise.transpose(1, 2, 0)), want=('err', 'q'))
    ref = c_oracle.closed_loop_batch(plan.q_start, noise, cfg['experiments']['desired_f'])
    assert np.array_equal(out['status'].cpu().numpy(), ref['status']) and np.array_equal(out['k_done'].cpu().numpy(), ref['k_done'])
    err = out['err'].cpu().numpy().transpose(2, 0, 1)
    dev = np.abs(err - ref['err']).max(axis=(1, 2)) / np.abs(ref['err']).max(axis=(1, 2))
    print(f'2048-trial batch vs C oracle, 299 steps: relative deviation of the error trajectories median {np.median(dev):.2e}, '
          f'99 % {np.quantile(dev, 0.99):.2e}, max {dev.max():.2e}')                                 # pytest -s shows it; DESIGN.md quotes it
    assert np.median(dev) <= 1e-11 and np.quantile(dev, 0.99) <= 1e-8 and dev.max() <= 1e-5          # contract: 1e-5 relative
    sdev = np.abs(out['stats'].cpu().numpy() - ref['stats']) / ref['stats']
    assert sdev.max() <= 1e-7


# ---------------------------------------------------------------------------------------------- full-size properties
def test_full_size_batch_properties(uvs):
    """BASELINE config 2 size (65 536 trials x 299 steps): results do not depend on batch position or launch
    partition, duplicates are bit-identical, sampled trials match the oracle, kappa/stat invariants hold."""
    import torch
    from oracle import plant_ref, rmckf_block, rmckf_dense
    g = load_golden('closed_gmckf_a1p5')
    meta = g['meta']
    T, K = 65536, 299
    fp = _fp(uvs, g)
    plant = uvs.SyntheticPlant.ur10(scene_desired(g))
    gen = torch.Generator(device='cuda').manual_seed(11)
    noise = torch.empty((K, 8, T), dtype=torch.float64, device='cuda')
    noise.cauchy_(generator=gen)                                             # impulsive, like alpha = 1
    noise.clamp_(-1e4, 1e4)
    q0 = torch.as_tensor(g['q_start'], device='cuda').repeat(T, 1)
    q0[:, 0] -= torch.rand(T, generator=gen, device='cuda', dtype=torch.float64) * 0.3
    q0[:, 1] -= torch.rand(T, generator=gen, device='cuda', dtype=torch.float64) * 0.6
    noise[:, :, 0] = torch.as_tensor(g['noise'], device='cuda')              # trial 0 = the reference fixture
    q0[0] = torch.as_tensor(g['q_start'], device='cuda')
    noise[:, :, T - 1] = noise[:, :, 12345]                                  # duplicate of another trial, far away in the grid
    q0[T - 1] = q0[12345]
    full = uvs.engine.closed_loop(fp, plant.to_struct(), q0, noise, want=('err',))
    err = full['err']
    assert rel_err(err[:, :, 0].cpu().numpy(), g['err']) <= 1e-8
    assert torch.equal(err[:, :, T - 1], err[:, :, 12345]) and torch.equal(full['stats'][T - 1], full['stats'][12345])
    # partition invariance: an odd-sized slice run on its own reproduces the same bits
    lo, hi = 30001, 30001 + 4097
    part = uvs.engine.closed_loop(fp, plant.to_struct(), q0[lo:hi].contiguous(), noise[:, :, lo:hi].contiguous(), want=('err',))
    assert torch.equal(part['err'], err[:, :, lo:hi]) and torch.equal(part['stats'], full['stats'][lo:hi])
    assert torch.equal(part['status'], full['status'][lo:hi]) and torch.equal(part['k_done'], full['k_done'][lo:hi])
    # stats kernel == fused stats
    ok = full['status'] == 0
    s2 = uvs.engine.stats_reduce(err, uvs.engine.loop_clock(meta['dt'], meta['t_max']), full['k_done'])
    assert torch.allclose(s2[ok], full['stats'][ok], rtol=1e-12, atol=0)
    # sampled trials against the oracle (short horizon: heavy-tailed closed loops amplify rounding later on)
    discs = plant_ref.place_discs()
    H = 25
    for t in (7, 4242, 65000):
        robot = plant_ref.PinholeUR10(meta['dt'])
        robot.start(q0[t].cpu().numpy())
        x0 = rmckf_dense.analytic_initial_guess(robot, robot.features(), 8, 6)
        ref = rmckf_block.run_closed_loop(lambda qq: plant_ref.project(plant_ref.fkine_all(qq)[5], discs), q0[t].cpu().numpy(),
                                          g['desired'], noise[:, :, t].cpu().numpy(), meta['dt'], meta['dt'] * (H + 0.5), meta['gain'], x0)
        assert rel_err(err[:H, :, t].cpu().numpy(), ref['err']) <= 1e-7
    assert int((full['status'] != 0).sum()) < T // 100


@pytest.mark.parametrize('method,max_failed,min_calm', [('GMCKF', 65, 0.975), ('MCKF', 1000, 0.98), ('KF', 65, 0.975), ('IMCCKF', 65, 0.985)])
def test_full_size_config2_product_noise_all_299_steps(uvs, method, max_failed, min_calm):
    """BASELINE config 2 exactly as bench.py runs it (VERDICT r3 #7a), and (round 5) the same launch for the other three estimators -- MCKF
    as the library's 8 tapered work items per trial at full size: 65 536 trials on the PRODUCT's alpha = 1.5 generator, global seeds
    123456 + t and jitter draws (main.py:121-139), noise through the shared T + 70-stream buffer (round 5).  536 trials spread over the grid
    (every 128th and the edges of wavefronts / rounds; round 4 sampled 24) are compared with oracle/c over ALL 299 steps on host noise of the
    same global indices (NoiseProfiler streams; the device generator matches them to 2e-13).  Heavy tails make a few closed loops amplify
    rounding: such trials are identified by the oracle itself, re-run from starts moved by 1e-14, and held to status /
    k_done only; the calm fraction is printed and gated per estimator at what was observed minus a small margin (``min_calm``)."""
    import torch
    import bench
    from oracle import c_oracle
    T, K = 65536, 299
    cfg = bench.config2()
    cfg['estimator']['method'] = method
    res = uvs.batch.run_batch(cfg, cells=[1.5], want=('err', 'q'))
    assert res.stats.shape == (T, 3) and len(res.plan) == T and int(res.plan.seed[0]) == 123456
    sample = np.unique(np.concatenate([np.arange(0, T, 128), [0, 1, 31, 32, 63, 64, 1000, 4097, 8191, 12345, 16384, 20000, 30001, 32767, 32768, 40000, 44444, 50000, 54321, 60000, 65000, 65503, 65534, 65535]]))
    noise = np.zeros((len(sample), K, 8))
    for i, t in enumerate(sample):
        uvs.batch.trial_noise(cfg, res.plan, int(t), int(t) + 1, K, noise[i:i + 1])
    des = cfg['experiments']['desired_f']
    ref = c_oracle.closed_loop_batch(res.plan.q_start[sample], noise, des, method=method)
    ref2 = c_oracle.closed_loop_batch(res.plan.q_start[sample] * (1.0 + 1e-14), noise, des, method=method)
    status, k_done = res.status.cpu().numpy(), res.k_done.cpu().numpy()
    err, q, stats = res.streams['err'], res.streams['q'], res.stats.cpu().numpy()
    calm = 0
    for i, t in enumerate(sample):
        same_verdict = int(ref['status'][i]) == int(ref2['status'][i]) and int(ref['k_done'][i]) == int(ref2['k_done'][i])
        if same_verdict:                                                     # (an MCKF trial near its subnormal-weight FAIL can end at another step
            assert int(ref['status'][i]) == int(status[t]) and int(ref['k_done'][i]) == int(k_done[t]), int(t)   # in the oracle's own re-run)
        kd = min(int(k_done[t]), int(ref['k_done'][i]), int(ref2['k_done'][i]))
        sens = max(rel_err(ref2['err'][i, :kd], ref['err'][i, :kd]), rel_err(ref2['q'][i, :kd], ref['q'][i, :kd]))
        if sens > 1e-11 or not same_verdict:
            continue                                                         # the oracle does not reproduce itself here: not a parity statement
        calm += 1
        assert rel_err(err[:kd, :, int(t)].cpu().numpy(), ref['err'][i, :kd]) <= 1e-8, int(t)
        assert rel_err(q[:kd, :, int(t)].cpu().numpy(), ref['q'][i, :kd]) <= 1e-8, int(t)
        if status[t] == 0:
            assert np.abs(stats[t] - ref['stats'][i]).max() / ref['stats'][i].max() <= 1e-8, int(t)
    print(f'full-size config 2, {method}: {calm} of {len(sample)} sampled trials calm ({calm / len(sample):.4f}) and within 1e-8 over all 299 steps; '
          f'{int((status != 0).sum())} of {T} trials FAILed')                    # pytest -s / the GPUTEST tail; DESIGN.md section 2 quotes it
    assert len(sample) >= 530 and calm >= min_calm * len(sample), (calm, len(sample))
    # the launch that bench.py times is this one: same failed-trial count as the bench line reports (headline 0; MCKF 581: the reference's
    # subnormal-weight path)
    assert int((status != 0).sum()) <= max_failed


def test_full_size_config3_properties(uvs):
    """BASELINE config 3 at its full size (262 144 trials x 299 steps, Gaussian mixture rho = 0.1 / mean 50 from the device generator,
    annealed sigma): partition invariance and duplicates bit-exact, fused statistics = statistics kernel, sampled trials against the
    block oracle on the same noise."""
    import torch
    import bench
    from oracle import plant_ref, rmckf_block, rmckf_dense
    T, K = 262144, 299
    cfg = bench.config2()
    cfg['noise'].update(type='GAUSSIAN_MIXTURE', noise_params={'std': 1.0, 'mean': 50.0, 'rho': 0.1}, hold=False, hold_time=0.5)
    cfg['estimator']['estimator_params']['annealing'] = True
    cfg['experiments']['epoch'] = T
    plan = uvs.batch.plan_trials(cfg, cells=[0.1])
    noise = uvs.batch.device_noise(cfg, plan, 0, T, K, 'cuda')
    q0 = torch.as_tensor(plan.q_start.copy(), device='cuda')
    noise[:, :, T - 1] = noise[:, :, 54321]                                 # a duplicate far away in the grid
    q0[T - 1] = q0[54321]
    des = cfg['experiments']['desired_f']
    fp = uvs.engine.make_params(8, 6, 'GMCKF', 10, True, 0.05, 15, 0.2, des, True, 0)
    plant = uvs.SyntheticPlant.ur10(des)
    full = uvs.engine.closed_loop(fp, plant.to_struct(), q0, noise, want=('err',))
    err = full['err']
    assert int(full['status'].sum()) == 0 and bool((full['k_done'] == K).all())
    assert torch.equal(err[:, :, T - 1], err[:, :, 54321]) and torch.equal(full['stats'][T - 1], full['stats'][54321])
    lo, hi = 200003, 200003 + 5001
    part = uvs.engine.closed_loop(fp, plant.to_struct(), q0[lo:hi].contiguous(), noise[:, :, lo:hi].contiguous(), want=('err',))
    assert torch.equal(part['err'], err[:, :, lo:hi]) and torch.equal(part['stats'], full['stats'][lo:hi])
    s2 = uvs.engine.stats_reduce(err, uvs.engine.loop_clock(0.05, 15), full['k_done'])
    assert torch.allclose(s2, full['stats'], rtol=1e-12, atol=0)
    discs = plant_ref.place_discs()
    for t in (7, 131071, 262143 - 1):
        robot = plant_ref.PinholeUR10(0.05)
        robot.start(plan.q_start[t])
        x0 = rmckf_dense.analytic_initial_guess(robot, robot.features(), 8, 6)
        nz = noise[:, :, t].cpu().numpy()
        ref = rmckf_block.run_closed_loop(lambda qq: plant_ref.project(plant_ref.fkine_all(qq)[5], discs), plan.q_start[t], des, nz, 0.05, 15, 0.2, x0,
                                          kernel_bw=10.0, annealing=True)
        assert ref['k_done'] == K and rel_err(err[:, :, t].cpu().numpy(), ref['err']) <= 1e-8


def test_full_size_config3_hold_properties(uvs):
    """BASELINE config 3 with the outlier hold ON (noise.py:103-116: a sample beyond 20 is repeated for 10 steps) at its full size,
    262 144 trials x 299 steps, noise from the device generator.  Held outliers make the closed loop amplify rounding (1.16x per step,
    DESIGN.md), so the pointwise closed-loop gate is a 40-step prefix; the strict gate over all 299 steps is the open-loop replay of
    the kernel's OWN f / dq streams through the block oracle.  Size-independent properties as for hold off."""
    import torch
    import bench
    from oracle import noise_ref, plant_ref, rmckf_block, rmckf_dense
    T, K = 262144, 299
    cfg = bench.config2()
    cfg['noise'].update(type='GAUSSIAN_MIXTURE', noise_params={'std': 1.0, 'mean': 50.0, 'rho': 0.1}, hold=True, hold_time=0.5)
    cfg['estimator']['estimator_params']['annealing'] = True
    cfg['experiments']['epoch'] = T
    plan = uvs.batch.plan_trials(cfg, cells=[0.1])
    noise = uvs.batch.device_noise(cfg, plan, 0, T, K, 'cuda')
    q0 = torch.as_tensor(plan.q_start.copy(), device='cuda')
    noise[:, :, T - 1] = noise[:, :, 54321]
    q0[T - 1] = q0[54321]
    des = cfg['experiments']['desired_f']
    fp = uvs.engine.make_params(8, 6, 'GMCKF', 10, True, 0.05, 15, 0.2, des, True, 0)
    plant = uvs.SyntheticPlant.ur10(des)
    full = uvs.engine.closed_loop(fp, plant.to_struct(), q0, noise, want=('err',))
    err = full['err']
    assert int(full['status'].sum()) == 0 and bool((full['k_done'] == K).all())
    assert torch.equal(err[:, :, T - 1], err[:, :, 54321]) and torch.equal(full['stats'][T - 1], full['stats'][54321])
    lo, hi = 200003, 200003 + 5001
    part = uvs.engine.closed_loop(fp, plant.to_struct(), q0[lo:hi].contiguous(), noise[:, :, lo:hi].contiguous(), want=('err', 'x', 'f', 'dq'))
    assert torch.equal(part['err'], err[:, :, lo:hi]) and torch.equal(part['stats'], full['stats'][lo:hi])
    s2 = uvs.engine.stats_reduce(err, uvs.engine.loop_clock(0.05, 15), full['k_done'])
    assert torch.allclose(s2, full['stats'], rtol=1e-12, atol=0)
    # the hold really holds: the generator's stream of a sampled trial equals the host restatement of noise.py, with runs of repeated outliers
    t = 131071
    ref_noise = noise_ref.NoiseStreamRef(8, noise_ref.GAUSSIAN_MIXTURE, int(plan.seed[t]), True, 10, std=1.0, mean=50.0, rho=0.1).take(K)
    got = noise[:, :, t].cpu().numpy()
    assert np.array_equal(got, ref_noise) and int((np.abs(got[1:]) > 20).sum()) > 50 and np.any((got[1:] == got[:-1]) & (np.abs(got[1:]) > 20))
    discs = plant_ref.place_discs()
    for t in (7, 131071, 262143 - 1):                                         # closed loop, 40-step prefix
        robot = plant_ref.PinholeUR10(0.05)
        robot.start(plan.q_start[t])
        x0 = rmckf_dense.analytic_initial_guess(robot, robot.features(), 8, 6)
        ref = rmckf_block.run_closed_loop(lambda qq: plant_ref.project(plant_ref.fkine_all(qq)[5], discs), plan.q_start[t], des, noise[:, :, t].cpu().numpy(), 0.05, 15,
                                          0.2, x0, kernel_bw=10.0, annealing=True)
        assert ref['k_done'] == K and rel_err(err[:40, :, t].cpu().numpy(), ref['err'][:40]) <= 1e-8
    for j in (0, 2500, 5000):                                                 # open loop over all 299 steps, trials lo + j
        robot = plant_ref.PinholeUR10(0.05)
        robot.start(plan.q_start[lo + j])
        f_init = robot.features()
        x0 = rmckf_dense.analytic_initial_guess(robot, f_init, 8, 6)
        f_seq = np.vstack([f_init[None], part['f'][:, :, j].cpu().numpy()])
        dq_prev = np.vstack([np.zeros((1, 6)), part['dq'][:-1, :, j].cpu().numpy()])
        ref = rmckf_block.run_replay(f_seq, dq_prev, x0, des, 0.2, 'GMCKF', 10.0, True, 300)
        assert rel_err(part['x'][:, :, j].cpu().numpy(), ref['X']) <= 1e-10
        assert rel_err(part['dq'][:, :, j].cpu().numpy(), ref['dq_cmd']) <= 1e-8


def test_full_size_config5_properties(uvs):
    """BASELINE config 5 at its full size (65 536 trials x 299 steps, (m, n) = (32, 7), per-trial records): partition invariance and
    duplicates bit-exact, fused statistics = statistics kernel, sampled trials against the block oracle."""
    import torch
    from oracle import rmckf_block
    T, K, M, N = 65536, 299, 32, 7
    lin = uvs.LinearPlant.random(M, N, seed=2)
    rng = np.random.default_rng(5)
    q_goal = lin.q0 + rng.uniform(-0.3, 0.3, N)
    des = lin.features(q_goal)
    q0_host = q_goal + np.random.default_rng(12345).uniform(-0.15, 0.15, (T, N))
    x0_row = (lin.J * (1 + 0.1 * rng.normal(size=lin.J.shape))).ravel()
    seeds = 123456 + np.arange(T)
    noise = uvs.noise_device.generate(uvs.NoiseType.ALPHA_STABLE, dict(alpha=1.5, beta=0, gamma=1, delta=0), seeds, M, K, layout='ktc', device='cuda')
    q0 = torch.as_tensor(q0_host, device='cuda')
    x0 = torch.as_tensor(np.tile(x0_row, (T, 1)), device='cuda')
    noise[:, T - 1, :] = noise[:, 4242, :]
    q0[T - 1] = q0[4242]
    fp = uvs.engine.make_params(M, N, 'GMCKF', 10, False, 0.05, 15, 0.2, des, False, 0)
    full = uvs.engine.closed_loop(fp, lin.to_struct(), q0, noise, x0, want=('err',), layout='ktc')
    err = full['err']                                                        # [step][trial][row]
    assert int(full['status'].sum()) == 0 and bool((full['k_done'] == K).all())
    assert torch.equal(err[:, T - 1], err[:, 4242]) and torch.equal(full['stats'][T - 1], full['stats'][4242])
    lo, hi = 40000, 40000 + 4096 + 5                                         # ragged: the last wavefront of the slice takes the plain store path
    part = uvs.engine.closed_loop(fp, lin.to_struct(), q0[lo:hi].contiguous(), noise[:, lo:hi].contiguous(), x0[lo:hi].contiguous(), want=('err',), layout='ktc')
    assert torch.equal(part['err'], err[:, lo:hi]) and torch.equal(part['stats'], full['stats'][lo:hi])
    s2 = uvs.engine.stats_reduce(err, uvs.engine.loop_clock(0.05, 15), full['k_done'], layout='ktc')
    assert torch.allclose(s2, full['stats'], rtol=1e-12, atol=0)
    for t in (3, 40001, 65534):
        ref = rmckf_block.run_closed_loop(lin.features, q0_host[t], des, noise[:, t].cpu().numpy(), 0.05, 15, 0.2, x0_row, initial_guess=False)
        assert ref['k_done'] == K and rel_err(err[:, t].cpu().numpy(), ref['err']) <= 1e-8


@pytest.mark.parametrize('method', ['GMCKF', 'KF', 'MCKF'])
def test_closed_loop_on_a_general_dh_table(uvs, method):
    """The tuned two-lane kernel picks an instantiation with compile-time zeros for UR10-like DH tables (alpha = -pi/2 on the first and 0 on
    the last link of either half chain).  A table that is NOT of that pattern -- every alpha tilted by a few hundredths of a radian -- must
    take the general chain code of the same kernel: 96 trials x 80 steps against oracle/c on the same tilted plant, and against the
    generic template; and on the UR10 table itself the two chain codes must agree to rounding."""
    from oracle import c_oracle
    import bench
    cfg = bench.config2()
    desired = np.array(cfg['experiments']['desired_f'])
    T, K = 96, 80
    rng = np.random.default_rng(31)
    q0 = np.tile(cfg['experiments']['q_start'], (T, 1)).astype(float)
    q0[:, :3] += rng.uniform(-0.15, 0.15, (T, 3))
    noise = 0.5 * rng.standard_t(3, size=(T, K, 8))
    tilted = uvs.SyntheticPlant.ur10(desired)
    tilted.alpha = tilted.alpha + np.array([0.02, -0.03, 0.015, -0.01, 0.025, 0.02])
    pl = c_oracle.ur10_plant()
    for i in range(6):
        pl.alpha[i] = tilted.alpha[i]
    for i, w in enumerate(tilted.points):
        for c in range(3):
            pl.points[i][c] = w[c]
    ref = c_oracle.closed_loop_batch(q0, noise, desired, method=method, steps=K, want_x=True, plant=pl)
    outs = {}
    for lanes in (0, -2):
        fp = uvs.engine.make_params(8, 6, method, 10.0, False, 0.05, 15.0, 0.2, desired, True, lanes, K)
        out = uvs.engine.closed_loop(fp, tilted.to_struct(), _cuda(q0), _cuda(noise.transpose(1, 2, 0)), want=('x', 'err', 'q'))
        assert np.array_equal(out['status'].cpu().numpy(), ref['status']) and np.array_equal(out['k_done'].cpu().numpy(), ref['k_done'])
        ok = ref['status'] == 0
        for key, rk in (('err', 'err'), ('q', 'q'), ('x', 'X')):
            assert rel_err(out[key].cpu().numpy().transpose(2, 0, 1)[ok], ref[rk][ok]) <= 1e-8, (lanes, key)
        outs[lanes] = out
    ur10 = uvs.SyntheticPlant.ur10(desired)
    nearly = uvs.SyntheticPlant.ur10(desired)
    nearly.alpha = nearly.alpha + np.array([1e-13, 0, 0, 0, 0, 0])           # sin(alpha_0) != -1 in the last bit or cos(alpha_0) > 1e-15: general code
    fp = uvs.engine.make_params(8, 6, method, 10.0, False, 0.05, 15.0, 0.2, desired, True, 0, K)
    a = uvs.engine.closed_loop(fp, ur10.to_struct(), _cuda(q0), _cuda(noise.transpose(1, 2, 0)), want=('err', 'q'))
    b = uvs.engine.closed_loop(fp, nearly.to_struct(), _cuda(q0), _cuda(noise.transpose(1, 2, 0)), want=('err', 'q'))
    assert rel_err(a['err'].cpu().numpy(), b['err'].cpu().numpy()) <= 1e-9 and rel_err(a['q'].cpu().numpy(), b['q'].cpu().numpy()) <= 1e-9


@pytest.mark.parametrize('method', ['GMCKF', 'KF', 'IMCCKF', 'MCKF'])
def test_pitched_streams_give_the_same_bits(uvs, method, monkeypatch):
    """Every entry point takes strides: input and output streams whose rows are pitched (UVS_ROW_PAD, engine.alloc_stream) -- an odd pad, so
    that nothing is 16-byte aligned any more -- give bit-identical results to dense rows, closed loop and replay."""
    import bench
    cfg = bench.config2()
    cfg['experiments']['epoch'] = 333
    plan = uvs.batch.plan_trials(cfg, cells=[1.5])
    K = 60
    fp = uvs.engine.make_params(8, 6, method, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0, steps=K, fpi_threshold=0.1, fpi_epoch_max=50)
    plant = uvs.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    q0 = _cuda(plan.q_start)
    want = ('x', 'err', 'q', 'f', 'dq')
    res = {}
    for pad in ('0', '37'):
        monkeypatch.setenv('UVS_ROW_PAD', pad)
        noise = uvs.batch.device_noise(cfg, plan, 0, len(plan), fp.steps, 'cuda', share=False)   # (every stream generated: dense or pitched rows)
        assert noise.is_contiguous() == (pad == '0')
        out = uvs.engine.closed_loop(fp, plant, q0, noise, want=want)
        assert out['x'].is_contiguous() == (pad == '0')
        rep = uvs.engine.replay(uvs.engine.make_params(8, 6, method, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], False, 0, steps=K - 1,
                                                       fpi_threshold=0.1, fpi_epoch_max=50),
                                out['f'], out['dq'][:K - 1], out['x'][0].permute(1, 0).contiguous(), want=('x', 'err'))
        res[pad] = {k: out[k].cpu().numpy() for k in want + ('stats', 'status', 'k_done')}
        res[pad].update({'rx': rep['x'].cpu().numpy(), 'rerr': rep['err'].cpu().numpy(), 'rstatus': rep['status'].cpu().numpy()})
    for k in res['0']:
        assert np.array_equal(res['0'][k], res['37'][k], equal_nan=True), k


@pytest.mark.parametrize('T', [333, 4096])
@pytest.mark.parametrize('method', ['GMCKF', 'KF', 'IMCCKF', 'MCKF'])
def test_record_layout_gives_the_same_bits(uvs, method, T):
    """with per-trial records for the streams ([step][trial][component], layout 'ktc') the (8,6) two-lane kernels of KF / IMCC-KF
    write X as whole 128-byte lines out of LDS (the XREC instantiation); RMCKF (measured: slower that way) and MCKF keep their strided stores.  Every stream, the statistics and
    the status bit for bit as with the trial-fastest layout -- ragged last wavefront included (T = 333: 13 of its 32 records exist)."""
    import torch
    import bench
    cfg = bench.config2()
    cfg['experiments']['epoch'] = T
    plan = uvs.batch.plan_trials(cfg, cells=[1.5])
    K = 45
    fp = uvs.engine.make_params(8, 6, method, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 2, steps=K, fpi_threshold=0.1, fpi_epoch_max=50)
    plant = uvs.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    q0 = _cuda(plan.q_start)
    noise = uvs.batch.device_noise(cfg, plan, 0, T, K, 'cuda', share=False)                # [K][8][T]
    want = ('x', 'err', 'q', 'f', 'dq')
    a = uvs.engine.closed_loop(fp, plant, q0, noise, want=want, layout='kct')
    b = uvs.engine.closed_loop(fp, plant, q0, noise.permute(0, 2, 1).contiguous(), want=want, layout='ktc')
    assert b['x'].shape == (K, T, 48) and b['x'].is_contiguous()
    assert torch.equal(a['status'], b['status']) and torch.equal(a['k_done'], b['k_done']) and torch.equal(a['stats'].view(torch.int64), b['stats'].view(torch.int64))
    for key in want:
        ta, tb = uvs.engine.as_tkc(a[key], 'kct'), uvs.engine.as_tkc(b[key], 'ktc')
        assert torch.equal(ta.contiguous().view(torch.int64), tb.contiguous().view(torch.int64)), key
    c = uvs.engine.closed_loop(fp, plant, q0, noise, want=want, layout='kct', x_layout='ktc')      # records for X alone, narrow streams trial-fastest
    assert c['x'].shape == (K, T, 48) and c['err'].shape == (K, 8, T) and torch.equal(c['x'].view(torch.int64), b['x'].view(torch.int64))
    assert torch.equal(c['err'].view(torch.int64), a['err'].view(torch.int64)) and torch.equal(c['stats'].view(torch.int64), a['stats'].view(torch.int64))


@pytest.mark.parametrize('lanes', [0, 8, 16])
@pytest.mark.parametrize('method', ['KF', 'IMCCKF', 'GMCKF'])
def test_wide_shape_records_equal_strided_streams(uvs, method, lanes):
    """(32,7) on the wide kernel: per-trial records (the LDS transposition; since round 5 KF / IMCC-KF hand their record stores out in pieces to the
    control law) against the trial-fastest layout, 8 and 16 lanes per filter, bit for bit.  tools/fuzz_long.py found the 16-lane KF kernel dropping its
    last records when the pieces were laid out for four rows per lane only."""
    import torch
    M, N, T, K = 32, 7, 40, 37
    lin = uvs.LinearPlant.random(M, N, seed=2)
    rng = np.random.default_rng(5)
    q_goal = lin.q0 + rng.uniform(-0.3, 0.3, N)
    q0 = _cuda(q_goal + rng.uniform(-0.15, 0.15, (T, N)))
    x0 = _cuda(np.tile((lin.J * (1 + 0.1 * rng.normal(size=lin.J.shape))).ravel(), (T, 1)))
    noise = rng.standard_t(3, size=(K, M, T)) * 0.3
    fp = uvs.engine.make_params(M, N, method, 10, True, 0.05, 15, 0.2, lin.features(q_goal), False, lanes, steps=K)
    plant = lin.to_struct('cuda')
    a = uvs.engine.closed_loop(fp, plant, q0, _cuda(noise), x0, want=('x', 'err', 'q'), layout='kct')
    b = uvs.engine.closed_loop(fp, plant, q0, _cuda(noise.transpose(0, 2, 1)), x0, want=('x', 'err', 'q'), layout='ktc')
    assert int(a['status'].sum()) == 0 and torch.equal(a['k_done'], b['k_done']) and torch.equal(a['stats'].view(torch.int64), b['stats'].view(torch.int64))
    for key in ('x', 'err', 'q'):
        assert torch.equal(uvs.engine.as_tkc(a[key], 'kct').contiguous().view(torch.int64), uvs.engine.as_tkc(b[key], 'ktc').contiguous().view(torch.int64)), key


def test_alloc_stream_row_pitch_knob(uvs, monkeypatch):
    """UVS_ROW_PAD pitches the rows of trial-fastest streams; the caller sees the [K][comp][T] view either way."""
    eng = uvs.engine
    monkeypatch.delenv('UVS_ROW_PAD', raising=False)
    dense = eng.alloc_stream(64, 5, 3, 'kct', 'cuda', zero=True)
    assert dense.shape == (5, 3, 64) and dense.is_contiguous()
    monkeypatch.setenv('UVS_ROW_PAD', '32')
    pitched = eng.alloc_stream(64, 5, 3, 'kct', 'cuda', zero=True)
    assert pitched.shape == (5, 3, 64) and pitched.stride() == (3 * 96, 96, 1) and not pitched.is_contiguous()
    assert eng.as_tkc(pitched, 'kct').shape == (64, 5, 3)
    v = eng.stream_view(pitched, 'kct')
    assert (v.trial_stride, v.step_stride, v.comp_stride) == (1, 3 * 96, 96)
    rec = eng.alloc_stream(64, 5, 3, 'ktc', 'cuda')                               # record layouts are never pitched
    assert rec.is_contiguous() and rec.shape == (5, 64, 3)


# ---------------------------------------------------------------------------------------------- UVS_OPT_LATENCY
def test_latency_option_picks_four_lanes_for_small_batches_and_stays_inside_the_gates(uvs):
    """the latency mapping (four lanes per filter, for shards that do not fill the chip) is an explicit option because the lane
    count changes the summation order of the least-squares reductions.  What it picks, that the reference fixture is still met at the same
    gate, and how far it moves a trial from the default mapping (measured here: last bits, amplified by the closed loop over 299 steps)."""
    import ctypes as C
    g = load_golden('closed_gmckf_a1p5')
    K = len(g['t'])
    plant = uvs.SyntheticPlant.ur10(scene_desired(g)).to_struct()
    lanes = lambda fp, T: int(uvs.lib().uvs_rmckf_closed_loop_lanes(C.byref(fp), C.byref(plant), T))      # noqa: E731
    fp = _fp(uvs, g)
    assert [lanes(fp, T) for T in (1, 8192, 16384, 16385, 65536)] == [4, 4, 4, 2, 2]      # by default too: the four-lane kernels with the two-lane bits
    fp.reserved = 2                                                           # UVS_OPT_LATENCY: the plain four-lane kernels
    assert [lanes(fp, T) for T in (1, 8192, 16384, 16385, 65536)] == [4, 4, 4, 2, 2]
    fp_m = _fp(uvs, g); fp_m.method = 3; fp_m.reserved = 2
    assert lanes(fp_m, 100) == 2                                              # MCKF keeps its two-lane kernel
    fp4 = _fp(uvs, g, 4); fp4.reserved = 2
    assert lanes(fp4, 65536) == 4 and lanes(_fp(uvs, g, -2), 100) == 2         # an explicit lanes_per_filter is not overridden
    T = 200
    rng = np.random.default_rng(4)
    q0 = np.tile(g['q_start'], (T, 1)); q0[1:, :2] -= rng.uniform(0, 0.3, (T - 1, 2))
    noise = np.repeat(g['noise'][:, :, None], T, axis=2); noise[:, :, 1:] = rng.standard_t(3, size=(K, 8, T - 1))
    outs = []
    for bits in (0, 2):
        fp = _fp(uvs, g); fp.reserved = bits
        outs.append(uvs.engine.closed_loop(fp, plant, _cuda(q0), _cuda(noise), want=('x', 'err', 'q')))
        err, q, X = (outs[-1][k].cpu().numpy()[:, :, 0] for k in ('err', 'q', 'x'))
        assert rel_err(err, g['err']) <= 1e-8 and rel_err(q, g['q']) <= 1e-8 and rel_err(X[g['X_steps']], g['X']) <= 1e-8
        assert int(outs[-1]['status'].sum()) == 0
    a, b = (o['err'].cpu().numpy() for o in outs)
    dev = np.abs(a - b).max(axis=(0, 1)) / np.abs(a).max(axis=(0, 1))
    assert not np.array_equal(a, b) and np.median(dev) <= 1e-12 and dev.max() <= 1e-8, (np.median(dev), dev.max())
    forced = uvs.engine.closed_loop(_fp(uvs, g, 4), plant, _cuda(q0), _cuda(noise), want=('err',))
    assert np.array_equal(forced['err'].cpu().numpy(), b)                     # the option = lanes_per_filter 4 at this size, bit for bit


@pytest.mark.parametrize('method', ['GMCKF', 'KF', 'IMCCKF'])
def test_small_batches_run_on_four_lanes_with_the_two_lane_bits(uvs, method):
    """VERDICT r3 #3a, lane-count-invariant arithmetic: up to 16 384 trials the library runs four lanes per filter (EMU2 kernels), which form every
    sum in the two-lane kernel's order -- so the choice is invisible: every stream, statistic and final state equals the two-lane kernel's
    (lanes_per_filter = 2) BIT FOR BIT.  Heavy-tailed noise, jittered starts, ragged batch sizes, annealing on and off, both DH code paths
    (the UR10 table's compile-time zeros and a general table), a non-finite sample that FAILs a trial."""
    import torch
    g = load_golden('closed_gmckf_a1p5')
    K = 120
    for T, anneal, tilt in ((1, False, False), (23, True, False), (200, False, False), (77, True, True)):
        rng = np.random.default_rng(T)
        plant = uvs.SyntheticPlant.ur10(scene_desired(g))
        if tilt:
            plant.alpha = np.asarray(plant.alpha, float).copy()
            plant.alpha[2] += 0.05                                            # no longer axis-aligned: the general chain code
        q0 = np.tile(g['q_start'], (T, 1)); q0[:, :3] += rng.uniform(-0.2, 0.1, (T, 3))
        noise = rng.standard_t(1.5, size=(K, 8, T)) * rng.choice([0.3, 1.0, 4.0], size=T)
        if T > 50:
            noise[40, 3, 17] = np.inf
        outs = []
        for lanes in (2, 0):
            fp = uvs.engine.make_params(8, 6, method, 10.0, anneal, 0.05, 15.0, 0.2, g['desired'], True, lanes, K)
            outs.append(uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0), _cuda(noise), want=('x', 'err', 'q', 'f', 'dq'), final_state=True))
        two, four = outs
        import ctypes as C
        assert int(uvs.lib().uvs_rmckf_closed_loop_lanes(C.byref(fp), C.byref(plant.to_struct()), T)) == 4
        assert torch.equal(two['status'], four['status']) and torch.equal(two['k_done'], four['k_done'])
        live = torch.arange(K, device='cuda')[:, None, None] < two['k_done'][None, None, :]
        for key in ('x', 'err', 'q', 'f', 'dq'):
            assert torch.equal(torch.where(live, two[key], 0.0).view(torch.int64), torch.where(live, four[key], 0.0).view(torch.int64)), (key, T)
        ok = two['status'] == 0
        for key in ('stats', 'x_final', 'p_final'):
            assert torch.equal(two[key][ok].view(torch.int64), four[key][ok].view(torch.int64)), (key, T)
        if T > 50:
            assert int(two['status'][17]) == 1 and int(two['k_done'][17]) == 40


@pytest.mark.parametrize('T', [8192, 16384])
def test_small_batch_kernels_at_the_shard_sizes_of_the_strong_series(uvs, T):
    """The shards a rank of north_star's 65 536-trial series owns at N = 8 and N = 4, exactly as bench.py runs them (config 2, product generator,
    global seeds and jitter, all 299 steps): the library's choice (four lanes per filter with the two-lane arithmetic) against two lanes forced --
    every logged value of every trial bit for bit, for the three estimators that have the small-batch kernels.  This is what keeps an N-GPU sweep
    bit-identical to the 1-GPU sweep although its shards run on another kernel."""
    import ctypes as C
    import torch
    import bench
    K = 299
    cfg = bench.config2()
    cfg['experiments']['epoch'] = 65536
    plan = uvs.batch.plan_trials(cfg, cells=[1.5])
    lo = 65536 - T                                                            # the LAST rank's shard: global trial numbers matter
    noise = uvs.batch.device_noise(cfg, plan, lo, 65536, K, 'cuda')
    q0 = torch.as_tensor(plan.q_start[lo:].copy(), device='cuda')
    des = cfg['experiments']['desired_f']
    plant = uvs.SyntheticPlant.ur10(des).to_struct()
    for method in ('GMCKF', 'KF', 'IMCCKF'):
        outs = []
        for lanes in (2, 0):
            fp = uvs.engine.make_params(8, 6, method, 10, False, 0.05, 15, 0.2, des, True, lanes)
            assert int(uvs.lib().uvs_rmckf_closed_loop_lanes(C.byref(fp), C.byref(plant), T)) == (2 if lanes else 4)
            outs.append(uvs.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err')))
        two, four = outs
        assert torch.equal(two['status'], four['status']) and torch.equal(two['k_done'], four['k_done']) and int(two['status'].sum()) == 0
        for key in ('x', 'err', 'stats'):
            assert torch.equal(two[key].view(torch.int64), four[key].view(torch.int64)), (method, key)
        del outs, two, four
        torch.cuda.empty_cache()


def test_segmented_rmckf_launch_is_bit_identical_to_whole_trials(uvs):
    """RMCKF launches that are not a whole number of rounds of wavefronts run their trials as four work items (the SEGMENTED instantiation of
    the two-lane kernel, uvs_rmckf_closed_loop_ws_f64): same arithmetic, state through the workspace -- every output equals the whole-trial
    launch bit for bit.  Forced on a small ragged batch (two lanes fixed: small batches would otherwise take the four-lane kernels) for 2 / 4 /
    7 segments, and once at a size the library cuts by itself (40 000 trials = 1.22 rounds)."""
    import ctypes as C
    import torch
    g = load_golden('closed_gmckf_a1p5')
    plant = uvs.SyntheticPlant.ur10(scene_desired(g)).to_struct()
    K, T = 120, 150
    rng = np.random.default_rng(9)
    q0 = np.tile(g['q_start'], (T, 1)); q0[:, :3] += rng.uniform(-0.2, 0.1, (T, 3))
    noise = rng.standard_t(1.5, size=(K, 8, T)) * rng.choice([0.3, 1.0, 4.0], size=T)
    noise[50, 2, 33] = np.nan                                                 # a trial that FAILs inside a segment
    outs = {}
    for n in (1, 2, 4, 7):
        fp = uvs.engine.make_params(8, 6, 'GMCKF', 10.0, True, 0.05, 15.0, 0.2, g['desired'], True, 2, K)
        fp.reserved = n << 8
        assert int(uvs.lib().uvs_rmckf_closed_loop_segments(C.byref(fp), C.byref(plant), T)) == n
        outs[n] = uvs.engine.closed_loop(fp, plant, _cuda(q0), _cuda(noise), want=('x', 'err', 'q', 'f', 'dq'), final_state=True)
    whole = outs[1]
    assert int(whole['status'][33]) == 1 and int(whole['k_done'][33]) == 50
    live = torch.arange(K, device='cuda')[:, None, None] < whole['k_done'][None, None, :]
    ok = whole['status'] == 0
    for n in (2, 4, 7):
        cut = outs[n]
        assert torch.equal(whole['status'], cut['status']) and torch.equal(whole['k_done'], cut['k_done'])
        for key in ('x', 'err', 'q', 'f', 'dq'):
            assert torch.equal(torch.where(live, whole[key], 0.0).view(torch.int64), torch.where(live, cut[key], 0.0).view(torch.int64)), (n, key)
        for key in ('stats', 'x_final', 'p_final'):
            assert torch.equal(whole[key][ok].view(torch.int64), cut[key][ok].view(torch.int64)), (n, key)
    # the library's own choice at 1.22 rounds
    import bench
    cfg = bench.config2()
    cfg['experiments']['epoch'] = 40000
    res = uvs.batch.run_batch(cfg, cells=[1.5], want=('err',))
    fp = uvs.engine.make_params(8, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0)
    assert int(uvs.lib().uvs_rmckf_closed_loop_segments(C.byref(fp), C.byref(plant), 40000)) == 4
    fp.reserved = 1 << 8
    whole = uvs.engine.closed_loop(fp, plant, torch.as_tensor(res.plan.q_start.copy(), device='cuda'), res.noise, want=('err',))
    assert torch.equal(whole['err'].view(torch.int64), res.streams['err'].view(torch.int64)) and torch.equal(whole['stats'].view(torch.int64), res.stats.view(torch.int64))

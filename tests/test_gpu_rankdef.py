"""numpy.linalg.pinv semantics of the control law (experiment.py:312: SVD, rcond = 1e-15) on the GPU path.

Fixtures `rankdef_*` come from the unmodified reference started on rank-deficient X0 (oracle/gen_golden_rankdef.py).  The kernels'
least squares flags such trials (|R_cc| spread) and the library's careful second pass redoes them with the SVD of the triangular factor;
what is tested: the result equals the reference's truncated minimum-norm command on every lane variant, healthy trials in the same batch
are not touched by the second pass, and no internal status leaks out."""
import numpy as np
import pytest

from conftest import RANKDEF_CMD_TOL, golden_names, load_golden, rel_err, scene_desired

pytestmark = pytest.mark.gpu

RANKDEF = golden_names('rankdef_')
HORIZON = {'rankdef_gmckf_dup_col': 40}          # identical columns separate by rounding; past pinv's cutoff the command is noise-driven
LANES_86 = (0, 1, 2, 4, -1, -2, -4, 8)


@pytest.fixture(scope='module')
def uvs():
    import uvs_amd
    uvs_amd.lib()
    return uvs_amd


def _cuda(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device='cuda')


def _fp(uvs, g, lanes=0, steps=None, strict=False):
    meta, p = g['meta'], g['meta']['params']
    fp = uvs.engine.make_params(8, 6, meta['method'], p['kernel_bw'], p['annealing'], meta['dt'], meta['t_max'], meta['gain'],
                                g['desired'], False, lanes, steps)
    fp.reserved = 1 if strict else 0                                                    # UVS_OPT_STRICT_PINV
    return fp


# The one fixture no magnitude of the triangular factor gives away (unit diagonal, every off-diagonal -1000: condition 3e18).  Until round 4
# the fast kernels returned the plain least-squares command there; since round 5 every solve also watches the GROWTH of its solution
# (max|sol| max|R| / max|Q^T y| >= 2^34, Spread::grows -- 2.4e15 or more on every step of this fixture, at most 4e2 on the healthy ones,
# tests/growth_watch_study.py), the trial is marked and the careful pass returns numpy's command in the default mode too.
KAHAN = 'rankdef_gmckf_kahan_c1000'


@pytest.mark.parametrize('lanes', LANES_86)
@pytest.mark.parametrize('name', RANKDEF)
def test_replay_matches_reference_on_rank_deficient_jacobians(uvs, name, lanes):
    g = load_golden(name)
    K = len(g['t'])
    f_seq = np.vstack([g['f_init'][None], g['f']])
    T = 3
    out = uvs.engine.replay(_fp(uvs, g, lanes), _cuda(np.repeat(f_seq[:, :, None], T, axis=2)), _cuda(np.repeat(g['dq_prev'][:, :, None], T, axis=2)),
                            _cuda(np.tile(g['X'][0], (T, 1))))
    h = HORIZON.get(name, K)
    X, cmd = out['x'].cpu().numpy(), out['dqcmd'].cpu().numpy()
    assert np.array_equal(cmd[:, :, 0], cmd[:, :, 2], equal_nan=True)
    assert rel_err(X[g['X_steps'], :, 0][:h], g['X'][:h]) <= 1e-10
    assert set(out['status'].cpu().numpy().tolist()) == {0} and int(out['k_done'][0]) == K
    assert rel_err(cmd[:h - 1, :, 0], g['dq_prev'][1:h]) <= RANKDEF_CMD_TOL.get(name, 1e-8)     # pinv's truncated minimum-norm command


@pytest.mark.parametrize('lanes', (0, 2, 4, -2))
@pytest.mark.parametrize('name', RANKDEF)
def test_strict_pinv_replay_matches_reference_everywhere(uvs, name, lanes):
    """UVS_OPT_STRICT_PINV: every solve through the careful kernels -- numpy's command on every fixture, the Kahan-like one included."""
    g = load_golden(name)
    K = len(g['t'])
    f_seq = np.vstack([g['f_init'][None], g['f']])
    T = 70
    out = uvs.engine.replay(_fp(uvs, g, lanes, strict=True), _cuda(np.repeat(f_seq[:, :, None], T, axis=2)), _cuda(np.repeat(g['dq_prev'][:, :, None], T, axis=2)),
                            _cuda(np.tile(g['X'][0], (T, 1))))
    h = HORIZON.get(name, K)
    X, cmd = out['x'].cpu().numpy(), out['dqcmd'].cpu().numpy()
    assert np.array_equal(cmd[:, :, 0], cmd[:, :, T - 1], equal_nan=True)
    assert rel_err(X[g['X_steps'], :, 0][:h], g['X'][:h]) <= 1e-10
    assert rel_err(cmd[:h - 1, :, 0], g['dq_prev'][1:h]) <= RANKDEF_CMD_TOL.get(name, 1e-8)
    assert set(out['status'].cpu().numpy().tolist()) == {0} and int(out['k_done'][0]) == K


@pytest.mark.parametrize('lanes,strict', [(0, False), (2, False), (4, False), (1, False), (-2, False), (0, True)])
def test_closed_loop_on_the_kahan_fixture(uvs, lanes, strict):
    """Closed loop from the Kahan-like X0: the reference's trajectory -- its truncated command is ~1e-5 rad/s, the plain
    least-squares one is hundreds of times larger -- in the DEFAULT mode on every lane mapping (the growth watch marks the trial at its first
    solve), and with UVS_OPT_STRICT_PINV as before."""
    g = load_golden(KAHAN)
    K = len(g['t'])
    plant = uvs.SyntheticPlant.ur10(scene_desired(g)).to_struct()
    args = (plant, _cuda(np.tile(g['q_start'], (3, 1))), _cuda(np.repeat(g['noise'][:, :, None], 3, axis=2)), _cuda(np.tile(g['X'][0], (3, 1))))
    out = uvs.engine.closed_loop(_fp(uvs, g, lanes, strict=strict), *args, want=('x', 'err', 'q', 'dq'))
    assert out['status'].cpu().tolist() == [0] * 3 and out['k_done'].cpu().tolist() == [K] * 3
    err, q, X, dq = (out[k].cpu().numpy()[:, :, 1] for k in ('err', 'q', 'x', 'dq'))
    assert rel_err(err, g['err']) <= 1e-7 and rel_err(q, g['q']) <= 1e-7 and rel_err(X[g['X_steps']], g['X']) <= 1e-7
    assert rel_err(dq[:K - 1], g['dq_prev'][1:]) <= 1e-6


def test_strict_pinv_agrees_with_the_fast_path_on_healthy_trials(uvs):
    """On a well-conditioned Jacobian the SVD finish equals the back substitution to rounding: the reference fixture within its gate on both."""
    g = load_golden('closed_gmckf_a1p5')
    K = len(g['t'])
    meta, p = g['meta'], g['meta']['params']
    plant = uvs.SyntheticPlant.ur10(scene_desired(g)).to_struct()
    outs = []
    for strict in (False, True):
        fp = uvs.engine.make_params(8, 6, 'GMCKF', p['kernel_bw'], p['annealing'], meta['dt'], meta['t_max'], meta['gain'], g['desired'], True, 0)
        fp.reserved = int(strict)
        outs.append(uvs.engine.closed_loop(fp, plant, _cuda(np.tile(g['q_start'], (40, 1))), _cuda(np.repeat(g['noise'][:, :, None], 40, axis=2)), want=('err', 'q')))
        assert rel_err(outs[-1]['err'].cpu().numpy()[:, :, 39], g['err']) <= 1e-8 and int(outs[-1]['status'].sum()) == 0
    assert rel_err(outs[1]['err'].cpu().numpy(), outs[0]['err'].cpu().numpy()) <= 1e-9


# closed loop: not the fixtures whose reference command is itself only defined to 1e-3 (the loop amplifies that), not the Kahan-like one (its own test above)
@pytest.mark.parametrize('lanes', (0, -2, 4))
@pytest.mark.parametrize('name', [n for n in RANKDEF if n not in HORIZON and n not in RANKDEF_CMD_TOL and n != KAHAN])
def test_closed_loop_matches_reference_on_rank_deficient_jacobians(uvs, name, lanes):
    g = load_golden(name)
    K = len(g['t'])
    plant = uvs.SyntheticPlant.ur10(scene_desired(g))
    out = uvs.engine.closed_loop(_fp(uvs, g, lanes), plant.to_struct(), _cuda(g['q_start'][None]), _cuda(g['noise'][:, :, None]),
                                 _cuda(g['X'][0][None]), want=('x', 'err', 'q', 'dq'))
    assert int(out['status'][0]) == 0 and int(out['k_done'][0]) == K
    err, q, X = (out[k].cpu().numpy()[:, :, 0] for k in ('err', 'q', 'x'))
    assert rel_err(err, g['err']) <= 1e-7 and rel_err(q, g['q']) <= 1e-7 and rel_err(X[g['X_steps']], g['X']) <= 1e-7
    from oracle.rmckf_dense import trial_stats
    assert rel_err(out['stats'].cpu().numpy()[0], trial_stats(g['err'], g['t'])) <= 1e-7


def test_second_pass_leaves_healthy_trials_alone(uvs):
    """A batch mixing healthy and rank-deficient X0: healthy trials come out bit-identical to a batch without the sick ones."""
    g, h = load_golden('rankdef_gmckf_rank4_product'), load_golden('closed_gmckf_a1p5')
    K, T = 120, 160
    sick = [3, 40]
    plant = uvs.SyntheticPlant.ur10(scene_desired(g))
    rng = np.random.default_rng(5)
    x0 = np.tile(h['X'][0], (T, 1)) * (1 + 0.02 * rng.standard_normal((T, 1)))
    noise = rng.standard_t(3, size=(K, 8, T))
    for t in sick:
        x0[t] = g['X'][0]
        noise[:, :, t] = g['noise'][:K]
    q0 = np.tile(g['q_start'], (T, 1))
    fp = _fp(uvs, g, 0, steps=K)
    out = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0), _cuda(noise), _cuda(x0), want=('x', 'err', 'q'))
    keep = [t for t in range(T) if t not in sick]
    ref = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0[keep]), _cuda(noise[:, :, keep]), _cuda(x0[keep]), want=('x', 'err', 'q'))
    # not bit-identical by design: a sick trial's first-pass joints run away, and one out-of-range angle switches its whole wavefront
    # from the bounded sincos to the library routine (rmckf_tuned.hpp), which rounds the neighbours' last bit differently
    for key in ('x', 'err', 'q'):
        assert rel_err(out[key].cpu().numpy()[:, :, keep], ref[key].cpu().numpy()) <= 1e-11
    assert rel_err(out['stats'].cpu().numpy()[keep], ref['stats'].cpu().numpy()) <= 1e-11
    far = [t for t in keep if t >= 64]                          # wavefronts (32 trials each) without a sick trial: bit-identical
    assert np.array_equal(out['err'].cpu().numpy()[:, :, far], ref['err'].cpu().numpy()[:, :, [keep.index(t) for t in far]])
    assert not out['status'].cpu().numpy().any() and not ref['status'].cpu().numpy().any()
    for t in sick:
        assert rel_err(out['err'].cpu().numpy()[:, :, t], g['err'][:K]) <= 1e-7 and rel_err(out['q'].cpu().numpy()[:, :, t], g['q'][:K]) <= 1e-7


def test_single_step_bank_uses_pinv_semantics(uvs):
    """External-robot route (FilterBank / uvs_rmckf_step_f64): the careful solve is inline."""
    g = load_golden('rankdef_gmckf_zero_and_scaled_col')
    fp = _fp(uvs, g, 0, steps=0)
    bank = uvs.engine.FilterBank(fp, 1, g['X'][0])
    f_seq = np.vstack([g['f_init'][None], g['f']])
    for k in range(25):
        dq, err, kap, st = bank.step(_cuda(f_seq[k + 1][None]), _cuda(f_seq[k][None]), _cuda(g['dq_prev'][k][None]), k)
        assert int(st[0]) == 0
        if k + 1 < 25:
            ref = g['dq_prev'][k + 1]
            assert np.abs(dq[0].cpu().numpy() - ref).max() <= 1e-8 * max(1e-3, np.abs(ref).max())


@pytest.mark.parametrize('lanes', (0, 16, -16))
def test_wide_shape_closed_loop_on_a_kahan_like_jacobian(uvs, lanes):
    """the (32,7) kernels solve the control law by the normal equations, whose Cholesky pivots -- like the |R_cc| of the QR
    -- say nothing about a Kahan-like Jacobian (unit-diagonal triangle, every off-diagonal -1000, condition 1e19), and whose solution does not
    even grow (squaring J has destroyed the small singular value).  Round 6 watches the REFINEMENT step instead: a correction of 2^-20 of the
    solution marks the trial for the careful pass (tests/growth_watch_study.py --wide: <= 7e-14 healthy, >= 2e-3 or a broken factorisation on
    Kahan-like steps).  Default mode, linear plant of BASELINE config 5, against the block oracle (numpy's pinv): the sick trial follows numpy's
    truncated command (4e-4 at step 0; the plain normal-equation command is 0.27), its healthy neighbours in the batch stay within 1e-8."""
    from oracle import rmckf_block
    plant = uvs.LinearPlant.random(32, 7, seed=2)
    rng = np.random.default_rng(77)
    q_goal = plant.q0 + rng.uniform(-0.3, 0.3, 7)
    des = plant.features(q_goal)
    T, K, sick = 5, 40, 2
    q0 = q_goal + rng.uniform(-0.15, 0.15, (T, 7))
    noise = rng.standard_t(3, size=(T, K, 32)) * 0.5
    x0 = np.tile((plant.J * (1 + 0.1 * rng.normal(size=plant.J.shape))).ravel(), (T, 1))
    qq, _ = np.linalg.qr(rng.normal(size=(32, 7)))
    kahan = 50.0 * qq @ (np.eye(7) - 1000.0 * np.triu(np.ones((7, 7)), 1))
    x0[sick] = kahan.ravel()
    y0 = plant.features(q0[sick]) - des
    plain = np.linalg.solve(kahan.T @ kahan, kahan.T @ y0)
    assert np.linalg.cond(kahan) > 1e18 and np.abs(plain).max() > 100 * np.abs(np.linalg.pinv(kahan) @ y0).max()      # the test has teeth
    fp = uvs.engine.make_params(32, 7, 'GMCKF', 10.0, False, 0.05, 15, 0.2, des, False, lanes, steps=K)
    out = uvs.engine.closed_loop(fp, plant.to_struct(), _cuda(q0), _cuda(noise.transpose(1, 2, 0)), _cuda(x0), want=('x', 'err', 'q', 'dq'))
    assert not out['status'].cpu().numpy().any() and np.all(out['k_done'].cpu().numpy() == K)
    for t in range(T):
        ref = rmckf_block.run_closed_loop(plant.features, q0[t], des, noise[t], 0.05, 0.05 * (K + 0.5), 0.2, x0[t], method='GMCKF', initial_guess=False)
        tol = 1e-6 if t == sick else 1e-8
        assert rel_err(out['err'].cpu().numpy()[:, :, t], ref['err']) <= tol, t
        assert rel_err(out['q'].cpu().numpy()[:, :, t], ref['q']) <= tol, t
        assert rel_err(out['x'].cpu().numpy()[:, :, t], ref['X']) <= tol, t
    dq0 = out['dq'].cpu().numpy()[0, :, sick]
    assert np.abs(dq0).max() < 1e-3                                        # numpy's truncated command, not the plain one (0.05)


@pytest.mark.parametrize('method,alpha', [('GMCKF', 1.5), ('GMCKF', 1.0), ('MCKF', 1.0), ('KF', 1.0)])
def test_strict_pinv_audit_at_full_size(uvs, method, alpha):
    """The audit UVS_OPT_STRICT_PINV exists for, at BASELINE config 2's size.  Since round 6 strict mode CERTIFIES every
    control-law solve in the tuned kernels -- |R|_F |R^-1|_F < 2^42 proves that numpy's pinv (experiment.py:312) truncates nothing, so the
    least-squares command is pinv's -- and sends only what it cannot certify to the SVD pass.  65 536 trials x 299 solves: the strict launch
    returns the default launch's streams, statistics, status and k_done BIT FOR BIT (a trial the certificate marked would come back from the
    careful kernel with other last bits), i.e. the default mode's watches missed nothing the certificate sees; and it costs at most 2 x."""
    import torch
    import bench
    cfg = bench.config2()
    cfg['estimator']['method'] = method
    cfg['noise']['noise_params']['alpha'] = alpha
    plan = uvs.batch.plan_trials(cfg, cells=[alpha])
    T, K = len(plan), 299
    noise = uvs.batch.device_noise(cfg, plan, 0, T, K, 'cuda')
    q0 = _cuda(plan.q_start)
    plant = uvs.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    outs, ms = [], []
    for strict in (False, True):
        fp = uvs.engine.make_params(8, 6, method, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0)
        fp.reserved = (1 if strict else 0) | (1 << 8)                # whole trials in both (strict mode never cuts MCKF trials into work items)
        uvs.engine.closed_loop(fp, plant, q0, noise, want=('err',))
        out = uvs.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
        torch.cuda.synchronize()
        ms.append(out['events'][0].elapsed_time(out['events'][1]))
        outs.append(out)
    a, b = outs
    assert torch.equal(a['status'], b['status']) and torch.equal(a['k_done'], b['k_done'])
    live = torch.arange(K, device='cuda')[:, None, None] < a['k_done'][None, None, :]
    for key in ('x', 'err', 'q'):
        assert torch.equal(torch.where(live, a[key], 0.0).view(torch.int64), torch.where(live, b[key], 0.0).view(torch.int64)), key
    ok = a['status'] == 0
    assert torch.equal(a['stats'][ok].view(torch.int64), b['stats'][ok].view(torch.int64))
    print(f'strict-pinv audit, {method} alpha {alpha}: {T} trials bit-identical to the default mode; {ms[1]:.2f} ms against {ms[0]:.2f} ms ({ms[1] / ms[0]:.2f} x)')
    assert ms[1] <= 2.0 * ms[0]

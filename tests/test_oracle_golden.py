"""The oracle is pinned to the reference: every restatement under oracle/ must reproduce the
golden vectors that oracle/gen_golden.py captured from the unmodified reference
(experiment.py Experiment.run, noise.py NoiseProfiler.getNoise) in the build container."""
import numpy as np
import pytest

from conftest import RANKDEF_CMD_TOL, NOISE_KIND, golden_names, load_golden, rel_err
from oracle import noise_ref, plant_ref, rmckf_block, rmckf_dense

CLOSED = golden_names('closed_')
NOISE = golden_names('noise_')
# closed loops whose feedback amplifies rounding differences (1e-15 -> O(1) within the run):
CHAOTIC = {'closed_gmckf_mix_anneal_hold'}


def _noise_stream(meta, m=8):
    return noise_ref.NoiseStreamRef(m, NOISE_KIND[meta['noise_type']], meta['seed'], meta['hold'], meta['hold_cnt'],
                                    **meta['noise_params'])


def test_fixture_inventory():
    assert len(NOISE) >= 16 and len(CLOSED) >= 17


@pytest.mark.parametrize('name', NOISE)
def test_noise_restatement_bit_exact(name):
    g = load_golden(name)
    got = _noise_stream(g['meta'], g['meta']['m']).take(len(g['values']))
    assert np.array_equal(got, g['values'])


def test_noise_known_answers_survey_appendix_b():
    g = load_golden('noise_alpha1p5')
    assert g['values'][0, 0] == 0.6341795339174561 and g['values'][1, 0] == -1.1841354142340323
    j = load_golden('noise_uniform_jitter')['values'][0]
    assert j[0] == 0.22733602246716966 and j[1] == 0.6711037852493347


def test_noise_seed_aliasing():
    """trial s feature i+1 == trial s+10 feature i (noise.py:70 + main.py:139)."""
    a = noise_ref.NoiseStreamRef(8, noise_ref.ALPHA_STABLE, 500, alpha=1.5, beta=0, gamma=1, delta=0).take(20)
    b = noise_ref.NoiseStreamRef(8, noise_ref.ALPHA_STABLE, 510, alpha=1.5, beta=0, gamma=1, delta=0).take(20)
    assert np.array_equal(a[:, 1:], b[:, :-1])


@pytest.mark.parametrize('name', CLOSED)
def test_dense_restatement_reproduces_reference(name):
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    out = rmckf_dense.run_closed_loop(
        plant_ref.PinholeUR10(meta['dt']), g['q_start'], g['desired'], _noise_stream(meta).next, meta['dt'], meta['t_max'],
        meta['gain'], method=meta['method'], initial_guess=p['initial_guess'], kernel_bw=p['kernel_bw'],
        annealing=p['annealing'], fpi_threshold=p['fpi_threshold'], fpi_epoch_max=p['fpi_epoch_max'], capture=True)
    assert out['status'] == int(g['status']) and out['k_done'] == len(g['t'])
    assert np.array_equal(out['t'], g['t']) and np.array_equal(out['noise'], g['noise'])
    # same numpy/LAPACK build => same bits; tolerance leaves room for a different BLAS on the GPU box
    tol = 1e-6 if name in CHAOTIC else 1e-11
    for key, ref in (('err', g['err']), ('q', g['q']), ('f', g['f'])):
        assert rel_err(out[key], ref) <= tol, key
    assert rel_err(out['X'][g['X_steps']], g['X']) <= tol
    assert rel_err(out['Pblk'][g['P_steps']], g['P_blocks']) <= tol
    assert float(g['P_offblock_max']) == 0.0                      # P is exactly block diagonal
    assert np.array_equal(out['dq'][:-1], g['dq_prev'][1:]) or rel_err(out['dq'][:-1], g['dq_prev'][1:]) <= tol


@pytest.mark.parametrize('name', CLOSED)
def test_block_replay_matches_reference(name):
    """Open-loop replay of the reference's recorded streams through the per-row form: no feedback, tight gate."""
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    f_seq = np.vstack([g['f_init'][None], g['f']])
    out = rmckf_block.run_replay(f_seq, g['dq_prev'], g['X'][0], g['desired'], meta['gain'], method=meta['method'],
                                 kernel_bw=p['kernel_bw'], annealing=p['annealing'], k_max=int(meta['t_max'] / meta['dt']),
                                 fpi_threshold=p['fpi_threshold'], fpi_epoch_max=p['fpi_epoch_max'])
    assert rel_err(out['X'][g['X_steps']], g['X']) <= 1e-11
    assert np.array_equal(out['err'], g['err'])
    # commanded dq of step k is the regressor of step k+1
    assert rel_err(out['dq_cmd'][:-1], g['dq_prev'][1:]) <= 1e-9
    last = int(g['P_steps'][-1])
    if last == len(g['t']) - 1:
        assert rel_err(out['P_final'], g['P_blocks'][-1]) <= 1e-11


@pytest.mark.parametrize('name', [n for n in CLOSED if 'gmckf' in n or '_kf_' in n or 'imcckf' in n])
def test_block_closed_loop_matches_reference(name):
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    discs = plant_ref.place_discs()
    robot = plant_ref.PinholeUR10(meta['dt'])
    robot.start(g['q_start'])
    x0 = rmckf_dense.analytic_initial_guess(robot, robot.features(), 8, 6)
    assert np.array_equal(x0.ravel(), g['X'][0])
    out = rmckf_block.run_closed_loop(lambda q: plant_ref.project(plant_ref.fkine_all(q)[5], discs), g['q_start'], g['desired'],
                                      g['noise'], meta['dt'], meta['t_max'], meta['gain'], x0, method=meta['method'],
                                      kernel_bw=p['kernel_bw'], annealing=p['annealing'])
    assert out['status'] == 0 and out['k_done'] == len(g['t'])
    horizon = 40 if name in CHAOTIC else len(g['t'])
    assert rel_err(out['err'][:horizon], g['err'][:horizon]) <= 1e-9
    assert rel_err(out['q'][:horizon], g['q'][:horizon]) <= 1e-9


def test_stats_known_answers_survey_appendix_b():
    g = load_golden('closed_gmckf_a1p5')
    s = rmckf_dense.trial_stats(g['err'], g['t'])
    assert np.allclose(s, [56047.5204, 5198.02211, 29866.6195], rtol=1e-8)
    g = load_golden('closed_gmckf_a1p5_anneal')
    assert np.allclose(rmckf_dense.trial_stats(g['err'], g['t']), [51999.6695, 4960.54063, 28561.2268], rtol=1e-8)


def test_plant_goal_pose_projects_onto_desired():
    discs = plant_ref.place_discs()
    f = plant_ref.project(plant_ref.fkine_all(plant_ref.Q_GOAL)[5], discs)
    assert np.abs(f - plant_ref.DESIRED_F).max() < 1e-9
    assert np.abs(discs[:, 2]).max() < 1e-12                      # discs lie on the floor


# ---------------------------------------------------------------------------------------------- the reference's tests/*.py
SCRIPTS = golden_names('script_')


def test_script_fixture_inventory():
    assert SCRIPTS == ['script_kalman_1_as_committed', 'script_kalman_1_initial_guess', 'script_kalman_3', 'script_mckf_1', 'script_mckf_3']


@pytest.mark.parametrize('name', SCRIPTS)
def test_block_replay_matches_reference_scripts(name):
    """Streams recorded while the reference's own tests/*.py ran (oracle/gen_golden_scripts.py): (m, n) = (2, 6) and (6, 6), KF with
    P = (I - KH) P (kalman_*:126/132) and fixed-point MCKF, regressing on the camera-twist command.  BASELINE.json configs[0]."""
    g = load_golden(name)
    meta = g['meta']
    out = rmckf_block.run_replay(g['f'], g['dp_prev'], g['X0'], g['desired'], meta['gain'], method=meta['method'],
                                 kernel_bw=meta['kernel_bw'] or 10.0, fpi_threshold=meta['fpi_threshold'],
                                 fpi_epoch_max=max(meta['fpi_epoch_max'], 1))
    K, mask = meta['steps'], g['cmd_mask']
    assert len(out['X']) == K
    assert rel_err(out['X'][g['X_steps']], g['X']) <= 1e-12
    assert rel_err(out['dq_cmd'][:-1][:, mask], g['dp_prev'][1:][:, mask]) <= 1e-10      # command of step k = regressor of step k+1
    assert int(g['P_steps'][-1]) == K - 1 and rel_err(out['P_final'], g['P_blocks'][-1]) <= 1e-11
    assert float(g['P_offblock_max']) == 0.0
    if meta['method'] == 'MCKF':
        assert np.array_equal(out['fpi_iterations'], g['epochs']) and g['epochs'].max() >= 2


# ---------------------------------------------------------------------------------------------- rank-deficient Jacobians (pinv semantics)
RANKDEF = golden_names('rankdef_')
# identical columns: the deficiency decays as rounding separates them; once sigma_6 crosses pinv's 1e-15 cutoff the command is noise-driven
RANKDEF_HORIZON = {'rankdef_gmckf_dup_col': 40}


def test_rankdef_fixture_inventory():
    assert RANKDEF == ['rankdef_gmckf_dup_col', 'rankdef_gmckf_kahan_c1000', 'rankdef_gmckf_rank1', 'rankdef_gmckf_rank4_product',
                       'rankdef_gmckf_scaled_1e12_indep', 'rankdef_gmckf_scaled_1e12_par_1em9', 'rankdef_gmckf_scaled_1e6_par_1em12',
                       'rankdef_gmckf_zero_and_scaled_col', 'rankdef_kf_rank4_product']
    for name in RANKDEF:
        g = load_golden(name)
        s = np.linalg.svd(g['x0'].reshape(8, 6), compute_uv=False)
        assert not g['meta']['params']['initial_guess']
        if name == 'rankdef_gmckf_scaled_1e12_indep':                                   # bad scaling alone: numpy keeps every component
            assert 1e-14 < s[-1] / s[0] < 1e-12
        else:
            assert s[-1] <= 1e-15 * s[0]                                                # pinv truncates from the first step on


def test_boundary_fixtures_sit_where_the_rank_watch_is_blind_or_not():
    """what an unpivoted QR shows of the boundary Jacobians, step by step along the reference's own runs (recomputed here
    from the fixtures' X with numpy).  `diag` = spread of |R_cc| (all the kernels watched until round 3), `entry` = max |R_ij| / min |R_cc|
    (watched since round 4), gate 2^34; `trunc` = numpy's pinv drops a singular value."""
    gate = 2.0 ** 34
    seen = {}
    for name in ('scaled_1e12_par_1em9', 'scaled_1e6_par_1em12', 'scaled_1e12_indep', 'kahan_c1000'):
        g = load_golden('rankdef_gmckf_' + name)
        assert len(g['X_steps']) == 299
        trunc, diag, entry = [], [], []
        for x in g['X']:
            sv = np.linalg.svd(x.reshape(8, 6), compute_uv=False)
            r = np.abs(np.linalg.qr(x.reshape(8, 6), mode='r'))
            d = np.diag(r)
            trunc.append(bool((sv <= 1e-15 * sv.max()).any())), diag.append(d.max() / d.min() >= gate), entry.append(r.max() / d.min() >= gate)
        seen[name] = (int(np.sum(trunc)), int(np.sum(diag)), int(np.sum(entry)), bool(entry[0]))
    assert seen['scaled_1e12_par_1em9'] == (299, 0, 299, True)        # truncated at every step; the diagonal never shows it, the entries always
    t, d, e, first = seen['scaled_1e6_par_1em12']
    assert t == 299 and d == 0 and first and 20 <= e < 299              # caught at the first step (one mark redoes the whole trial), not at every step
    assert seen['scaled_1e12_indep'][0] == 0 and seen['scaled_1e12_indep'][2] == 299      # marked although numpy truncates nothing: the careful pass must agree anyway
    assert seen['kahan_c1000'] == (299, 0, 0, False)                    # no magnitude of the factor shows it: UVS_OPT_STRICT_PINV territory


@pytest.mark.parametrize('name', RANKDEF)
def test_block_replay_matches_reference_on_rank_deficient_jacobians(name):
    """The reference run from a rank-deficient X0 (oracle/gen_golden_rankdef.py): every command is pinv's truncated minimum-norm solution."""
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    f_seq = np.vstack([g['f_init'][None], g['f']])
    assert np.array_equal(g['X'][0], g['x0']) and not np.any(g['f_init'])
    out = rmckf_block.run_replay(f_seq, g['dq_prev'], g['X'][0], g['desired'], meta['gain'], method=meta['method'],
                                 kernel_bw=p['kernel_bw'], annealing=p['annealing'], k_max=int(meta['t_max'] / meta['dt']))
    h = RANKDEF_HORIZON.get(name, len(g['t']))
    assert rel_err(out['X'][g['X_steps']][:h], g['X'][:h]) <= 1e-11
    assert rel_err(out['dq_cmd'][:h - 1], g['dq_prev'][1:h]) <= RANKDEF_CMD_TOL.get(name, 1e-8)


def test_numpy_pinv_is_only_defined_to_cond_eps_on_the_1e12_fixtures():
    """Why two fixtures carry a loose command gate: one ulp on the reference's own X moves the reference's own command."""
    def sensitivity(name):
        g = load_golden(name)
        worst = 0.0
        for k in range(0, 299, 7):
            x, y = g['X'][k].reshape(8, 6), g['err'][k]
            a, b = np.linalg.pinv(x) @ y, np.linalg.pinv(np.nextafter(x, np.inf)) @ y
            worst = max(worst, float(np.abs(a - b).max() / np.abs(a).max()))
        return worst
    assert 1e-4 < sensitivity('rankdef_gmckf_scaled_1e12_indep') < RANKDEF_CMD_TOL['rankdef_gmckf_scaled_1e12_indep']
    assert 1e-5 < sensitivity('rankdef_gmckf_scaled_1e12_par_1em9') < RANKDEF_CMD_TOL['rankdef_gmckf_scaled_1e12_par_1em9']
    assert sensitivity('rankdef_gmckf_scaled_1e6_par_1em12') < 1e-8 and sensitivity('rankdef_gmckf_kahan_c1000') < 1e-12


# ---------------------------------------------------------------------------------------------- MCKF fixed-point iteration (fpi_*)
FPI = golden_names('fpi_')
FPI_HORIZON = {}            # none of these runs amplifies rounding on this plant (block vs dense over the whole trial: <= 1e-11)


def test_fpi_fixture_inventory():
    """oracle/gen_golden_fpi.py: runs of the unmodified reference that take the multi-pass, epoch-cap, Cy = 0 and subnormal-Cy paths."""
    assert FPI == ['fpi_default_a1p0_seed0', 'fpi_default_a1p0_seed38', 'fpi_mckf_a1p0_bw1_fail', 'fpi_mckf_a1p0_thr1em3', 'fpi_mckf_a1p2_cap3',
                   'fpi_mckf_a1p2_cap4', 'fpi_mckf_a1p2_thr1em6_fail', 'fpi_mckf_a1p5_anneal_thr1em4', 'fpi_mckf_a1p5_thr1em6']
    for name in FPI:
        g = load_golden(name)
        k, ep, sk = len(g['t']), g['fpi_epochs'], g['fpi_skip']
        failed = int(g['status']) == 1
        assert len(ep) == len(sk) == k + failed and g['noise_full'].shape == (299, 8) and np.array_equal(g['noise_full'][:k], g['noise'])
        assert failed == (name.endswith('_fail') or 'default' in name) and (k == 299) != failed
        if 'thr' in name or 'cap' in name:
            assert int((ep >= 2).sum()) >= 10, name                  # the second and later passes really run
        if 'cap' in name:
            cap = g['meta']['params']['fpi_epoch_max']
            assert ep.max() == cap and np.array_equal(sk[ep == cap], np.ones(int((ep == cap).sum()), bool)) and 10 <= int(sk.sum()) < k - 10
        if 'a1p0' in name:
            assert int((ep == 0).sum()) >= 2 and np.all(sk[ep == 0])   # a zero weight Cy: inv raises before the first pass completes
        if failed:
            assert not sk[-1] and ep[-1] == 1                          # subnormal Cy: one pass, NaN state, not a skipped correction


@pytest.mark.parametrize('name', FPI)
def test_dense_restatement_reproduces_reference_fpi(name):
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    it = iter(g['noise_full'])
    out = rmckf_dense.run_closed_loop(
        plant_ref.PinholeUR10(meta['dt']), g['q_start'], g['desired'], lambda: next(it), meta['dt'], meta['t_max'],
        meta['gain'], method='MCKF', initial_guess=True, kernel_bw=p['kernel_bw'], annealing=p['annealing'],
        fpi_threshold=p['fpi_threshold'], fpi_epoch_max=p['fpi_epoch_max'], capture=True)
    k = len(g['t'])
    assert out['status'] == int(g['status']) and out['k_done'] == k
    assert np.array_equal(out['fpi_iterations'], g['fpi_epochs']) and np.array_equal(out['fpi_skipped'], g['fpi_skip'])
    for key, ref in (('err', g['err']), ('q', g['q']), ('f', g['f'])):
        assert rel_err(out[key], ref) <= 1e-9, key
    assert rel_err(out['X'][g['X_steps']], g['X']) <= 1e-9


@pytest.mark.parametrize('name', FPI)
def test_block_replay_matches_reference_fpi(name):
    """The reference's recorded streams through the per-row fixed-point iteration: X per step, iteration counts, skipped corrections."""
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    k = len(g['t'])
    f_seq = np.vstack([g['f_init'][None], g['f']])
    out = rmckf_block.run_replay(f_seq, g['dq_prev'], g['X'][0], g['desired'], meta['gain'], method='MCKF', kernel_bw=p['kernel_bw'],
                                 annealing=p['annealing'], k_max=300, fpi_threshold=p['fpi_threshold'], fpi_epoch_max=p['fpi_epoch_max'])
    assert np.array_equal(out['fpi_iterations'], g['fpi_epochs'][:k])
    assert rel_err(out['X'][g['X_steps']], g['X']) <= 1e-10
    assert rel_err(out['dq_cmd'][:-1], g['dq_prev'][1:]) <= 1e-8
    if int(g['P_steps'][-1]) == k - 1:
        assert rel_err(out['P_final'], g['P_blocks'][-1]) <= 1e-10


@pytest.mark.parametrize('name', FPI)
def test_block_closed_loop_matches_reference_fpi(name):
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    discs = plant_ref.place_discs()
    out = rmckf_block.run_closed_loop(lambda q: plant_ref.project(plant_ref.fkine_all(q)[5], discs), g['q_start'], g['desired'],
                                      g['noise_full'], meta['dt'], meta['t_max'], meta['gain'], g['X'][0], method='MCKF',
                                      kernel_bw=p['kernel_bw'], annealing=p['annealing'], fpi_threshold=p['fpi_threshold'], fpi_epoch_max=p['fpi_epoch_max'])
    k = len(g['t'])
    h = min(k, FPI_HORIZON.get(name, k))
    assert rel_err(out['err'][:h], g['err'][:h]) <= 1e-8 and rel_err(out['q'][:h], g['q'][:h]) <= 1e-8
    assert np.array_equal(out['fpi_iterations'][:h], g['fpi_epochs'][:h])
    if name not in FPI_HORIZON:
        assert out['status'] == int(g['status']) and out['k_done'] == k

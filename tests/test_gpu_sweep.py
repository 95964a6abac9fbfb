"""batch.run_sweep against the reference's EXPERIMENT: the fourteen 1 200-trial Monte-Carlo tables the unmodified main.py produced in the build
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


@pytest.mark.parametrize('name', ['r2_gmckf', 'r2_mckf', 'r1_kf', 'r3_gmckf_anneal'])
def test_drop_in_classes_reproduce_rows_of_the_reference_experiment(name):
    """main.py:104-148 written against the PACKAGE's drop-in classes -- NoiseProfiler(seed0 + t), the jitter stream, Experiment(...).run() -- for one trial
    of every cell of the reference's experiment (the first of each cell: global trial index 100 c), reduced as results/plot_errorbar.m:39-84 reduces
    the reference's CSV: status, number of logged rows and ||ISE|| / ||IAE|| / ||ITAE|| equal the rows of the reference's own run (tests/golden/sweep_*.npz) --
    wherever the oracle says the trial is reproducible.  Both routes of Experiment: the whole trial in one kernel (SyntheticRobot), and -- first and last
    cell -- the per-step route (uvs_rmckf_step_f64 on pinned records) around a robot the package knows nothing about."""
    import uvs_amd as uvs
    from oracle.plant_ref import PinholeUR10
    from sweep_common import calm_mask
    ref = load_sweep(name)
    cfg = ref['config']
    ex, est, nz = cfg['experiments'], cfg['estimator'], cfg['noise']
    desired_f = np.array(ex['desired_f'])
    rho_list = np.linspace(1, 2, 12)                                     # main.py:104-106
    jitter = uvs.NoiseProfiler(num_features=2, noise_type=uvs.NoiseType.UNIFORM, seed=ex['seed']) if ex['change_q_start'] else None
    picked, rows = [100 * c for c in range(12)], {}
    t_global = 0
    for c, rho in enumerate(rho_list):
        for i in range(ex['epoch']):
            q = np.array(ex['q_start'], float)
            if jitter is not None:                                       # every trial consumes its draw (main.py:129-134)
                r = jitter.getNoise()
                q[0] = q[0] + 2 * (r[0] - 1) * (np.pi / 18)
                q[1] = q[1] + 2 * (r[1] - 1) * (np.pi / 9)
            if t_global in picked:
                for route in (('device_plant', 'external_robot') if c in (0, 11) else ('device_plant',)):
                    prof = uvs.NoiseProfiler(num_features=len(desired_f), noise_type=uvs.NoiseType[nz['type']], seed=nz['seed'] + t_global, noise_hold=nz['hold'],
                                             noise_hold_cnt=int(nz['hold_time'] / ex['dt']), noise_params=dict(nz['noise_params'], alpha=rho))
                    robot = uvs.SyntheticRobot(dt=ex['dt']) if route == 'device_plant' else PinholeUR10(ex['dt'])
                    run = uvs.Experiment(q_start=q, desired_f=desired_f, noise_prof=prof, t_s=ex['dt'], t_max=ex['t_max'], ibvs_gain=ex['ibvs_gain'], robot=robot,
                                         method=uvs.Method[est['method']], method_params=est['estimator_params']).run()
                    status, t_log, _, q_log, f_log, fd_log = run[:6]
                    e = fd_log - f_log                                   # plot_errorbar.m:39-46
                    stats = np.array([np.linalg.norm((e * e).sum(0)), np.linalg.norm(np.abs(e).sum(0)), np.linalg.norm(t_log @ np.abs(e))])
                    rows[(t_global, route)] = (status.value, len(t_log), stats, q_log[0] if len(q_log) else q)
            t_global += 1
    plan = uvs.batch.plan_trials(cfg)
    calm = dict(zip(picked, calm_mask(uvs, cfg, plan, np.array(picked))))
    checked = 0
    for (t, route), (status, k, stats, q_first) in rows.items():
        assert np.array_equal(q_first, ref['q_first'][t]), (t, route)                    # the start the reference's trial had, jitter included
        if not calm[t]:
            continue
        checked += 1
        assert status == int(ref['status'][t]) and k == int(ref['k_done'][t]), (name, t, route)
        assert np.abs(stats - ref['stats'][t]).max() <= 1e-8 * np.abs(ref['stats'][t]).max(), (name, t, route, stats, ref['stats'][t])
    print(f'{name}: {checked} of {len(rows)} Experiment.run() trials compared with the reference experiment\'s rows (the others are chaotic in the oracle)')
    assert checked >= len(rows) - 4

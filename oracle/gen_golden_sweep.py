"""Generate tests/golden/sweep_*.npz: the reference's EXPERIMENT -- the 12-cell x ``epoch`` Monte-Carlo table of main.py:104-196 at full
length (t_max 15, 299 logged steps), reduced per trial the way results/plot_errorbar.m:20-98 reduces results.csv.

BUILD-CONTAINER ONLY (imports /root/reference, read-only; only the npz tables travel).  Same ``runpy`` harness as gen_golden_csv.py: the
reference's UNMODIFIED main.py is executed in a scratch directory against the kinematic pinhole plant; it writes its own results.csv
(1 200 trials x 299 rows x 41 columns, ~290 MB of text per job), which is then read back and reduced:

  per trial   status (0 SUCCESS / 1 FAIL, from the ``ExperimentStatus.*`` string), k_done (rows the reference logged), rho (the swept
              value), ||ISE||_2, ||IAE||_2, ||ITAE||_2 over the 8 features with e_j = desired_f_j - f_j (plot_errorbar.m:39-84),
              the first logged joint vector (pins the q_start jitter of main.py:129-134) and the last logged feature row;
  per cell    number of SUCCESS trials (plot_errorbar.m:25 removes FAILs) and mean / std (N - 1, MATLAB's) / median of the three norms.

Jobs (``python oracle/gen_golden_sweep.py [job ...]``; none = all, four at a time):

  results1 protocol  ALPHA_STABLE, alpha = linspace(1, 2, 12), fixed q_start, epoch 100:  r1_kf  r1_mckf  r1_imcckf  r1_gmckf
  results2 protocol  the same with change_q_start (the shipped config.json), epoch 100:   r2_kf  r2_mckf  r2_imcckf  r2_gmckf
  results3 protocol  GMCKF: annealing / sigma = 1 (sigma = 10 is r1_gmckf), epoch 100:     r3_gmckf_anneal  r3_gmckf_sigma1
  mixture sweep      GAUSSIAN_MIXTURE rho = linspace(0, .2, 12), annealed sigma, hold off / on:  r4_gmckf_mix_anneal  r4_gmckf_mix_anneal_hold
  other noise laws   GAUSSIAN_BIMODAL under IMCC-KF, WHITE_NOISE under KF:                     r5_imcckf_bimodal  r5_kf_white

The noise seed schedule (seed0 + global trial index) and the jitter stream are main.py's own; nothing of the reference is patched except
the simulator class, the detector binding and the clock of the plant (gen_golden_csv.py).
"""
import glob
import json
import os
import runpy
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

#        name              method    change_q  annealing  kernel_bw  epoch
JOBS = {'r1_kf':           ('KF',     False,    False,     10,        100),
        'r1_mckf':         ('MCKF',   False,    False,     10,        100),
        'r1_imcckf':       ('IMCCKF', False,    False,     10,        100),
        'r1_gmckf':        ('GMCKF',  False,    False,     10,        100),
        'r2_kf':           ('KF',     True,     False,     10,        100),
        'r2_mckf':         ('MCKF',   True,     False,     10,        100),
        'r2_imcckf':       ('IMCCKF', True,     False,     10,        100),
        'r2_gmckf':        ('GMCKF',  True,     False,     10,        100),
        'r3_gmckf_anneal': ('GMCKF',  False,    True,      10,        100),
        'r3_gmckf_sigma1': ('GMCKF',  False,    False,     1,         100),
        # BASELINE config 3's noise at experiment level: GAUSSIAN_MIXTURE, rho = linspace(0, 0.2, 12) (main.py:104), std 1 / mean 50 (the reference ships no
        # defaults for them; the outliers must exceed the hold threshold of 20, noise.py:103), annealed sigma, outlier hold off / on (0.5 s = 10 steps)
        'r4_gmckf_mix_anneal':      ('GMCKF', True, True, 10, 100),
        'r4_gmckf_mix_anneal_hold': ('GMCKF', True, True, 10, 100),
        # the remaining noise laws at experiment level (main.py sweeps rho for every type but ALPHA_STABLE; WHITE_NOISE ignores it: twelve cells that differ in
        # their seeds only): the three-component bimodal mixture with its selector streams (noise.py:140-148) under IMCC-KF, white noise under KF
        'r5_imcckf_bimodal': ('IMCCKF', True, False, 10, 100),
        'r5_kf_white':       ('KF',     True, False, 10, 100)}


def job_config(name):
    method, change_q, anneal, bw, epoch = JOBS[name]
    cfg = json.load(open(os.path.join(OUT, 'config_reference.json')))
    cfg.pop('_provenance', None)
    cfg['log_level'] = 'CRITICAL'
    cfg['experiments'].update(epoch=epoch, change_q_start=change_q)
    cfg['estimator']['method'] = method
    cfg['estimator']['estimator_params'].update(annealing=anneal, kernel_bw=bw)
    if name.startswith('r4_'):
        cfg['noise'].update(type='GAUSSIAN_MIXTURE', noise_params={'std': 1.0, 'mean': 50.0}, hold=name.endswith('_hold'), hold_time=0.5)
    if name == 'r5_imcckf_bimodal':
        cfg['noise'].update(type='GAUSSIAN_BIMODAL', noise_params={'std': 1.0, 'mean': 30.0}, hold=False, hold_time=0.5)
    if name == 'r5_kf_white':
        cfg['noise'].update(type='WHITE_NOISE', noise_params={'std': 2.0}, hold=False, hold_time=0.5)
    return cfg


def reduce_csv(path):
    """plot_errorbar.m:20-98 on the reference's results.csv, plus the per-trial rows."""
    import pandas as pd
    df = pd.read_csv(path, float_precision='round_trip')             # the default parser is fast and up to an ulp off; repr text round-trips exactly
    ids = df['experiment_id'].to_numpy()
    T = int(ids.max()) + 1
    start = np.searchsorted(ids, np.arange(T))                          # rows of a trial are contiguous and in order (main.py:196)
    stop = np.searchsorted(ids, np.arange(T), side='right')
    assert np.all(np.diff(ids) >= 0) and np.all(stop > start)
    f = df[[f'f_{i}' for i in range(1, 9)]].to_numpy()
    fd = df[[f'desired_f_{i}' for i in range(1, 9)]].to_numpy()
    q = df[[f'q_{i}' for i in range(1, 7)]].to_numpy()
    t = df['t'].to_numpy()
    e = fd - f                                                          # plot_errorbar.m:39-46
    status_row = (df['status'].to_numpy() != 'ExperimentStatus.SUCCESS').astype(np.int32)
    assert set(df['status'].unique()) <= {'ExperimentStatus.SUCCESS', 'ExperimentStatus.FAIL'}
    rows = dict(status=np.zeros(T, np.int32), k_done=np.zeros(T, np.int32), rho=np.zeros(T), stats=np.zeros((T, 3)),
                q_first=np.zeros((T, 6)), f_last=np.zeros((T, 8)), t_last=np.zeros(T))
    for j in range(T):
        a, b = start[j], stop[j]
        ej, tj = e[a:b], t[a:b]
        ise = np.array([ej[:, c] @ ej[:, c] for c in range(8)])         # :49-58
        iae = np.array([np.abs(ej[:, c]).sum() for c in range(8)])      # :61-70
        itae = np.array([tj @ np.abs(ej[:, c]) for c in range(8)])      # :73-82
        rows['stats'][j] = np.linalg.norm(ise), np.linalg.norm(iae), np.linalg.norm(itae)
        rows['status'][j], rows['k_done'][j], rows['rho'][j] = status_row[a], b - a, df['rho'].iat[a]
        rows['q_first'][j], rows['f_last'][j], rows['t_last'][j] = q[a], f[b - 1], t[b - 1]
    rhos = np.unique(rows['rho'])                                       # :11
    cell = dict(cells=rhos, n_success=np.zeros(len(rhos), np.int32), n_fail=np.zeros(len(rhos), np.int32),
                mean=np.full((len(rhos), 3), np.nan), std=np.full((len(rhos), 3), np.nan), median=np.full((len(rhos), 3), np.nan))
    for i, r in enumerate(rhos):
        sel = (rows['rho'] == r)
        ok = sel & (rows['status'] == 0)                                # :25
        cell['n_success'][i], cell['n_fail'][i] = ok.sum(), sel.sum() - ok.sum()
        if ok.sum():
            s = rows['stats'][ok]
            cell['mean'][i], cell['median'][i] = s.mean(0), np.median(s, 0)
            cell['std'][i] = s.std(0, ddof=1) if ok.sum() > 1 else 0.0  # MATLAB std: N - 1
    return rows, cell


def quat_from_rotation(R):
    """Scalar-first unit quaternion of a rotation matrix, largest-component branch (trials under Cauchy noise turn the camera anywhere;
    gen_golden_csv.py's w > 0 branch is enough only for its 9-row trials).  Feeds the camera_* columns, which the reduction does not read."""
    d = np.array([R[0, 0] + R[1, 1] + R[2, 2], R[0, 0], R[1, 1], R[2, 2]])
    i = int(np.argmax(d))
    if i == 0:
        w = 0.5 * np.sqrt(max(1.0 + d[0], 0.0))
        return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
    a, b, c = i - 1, i % 3, (i + 1) % 3
    s = 0.5 * np.sqrt(max(1.0 + R[a, a] - R[b, b] - R[c, c], 0.0))
    q = np.zeros(4)
    q[0], q[1 + a], q[1 + b], q[1 + c] = (R[c, b] - R[b, c]) / (4 * s), s, (R[b, a] + R[a, b]) / (4 * s), (R[c, a] + R[a, c]) / (4 * s)
    return q


def run_job(name):
    sys.path.insert(0, HERE)
    import gen_golden as G                                              # stubs cv2 / zmq, imports the reference

    class SweepPlant(G.RefPlant):
        def __init__(self, logger=None, visualization=False):
            super().__init__()

        def computePose(self, recalculate_fkine=False):
            import utils                                                # reference
            T = self.fkine(recalculate=True)
            R = T[:3, :3]
            return np.r_[T[:3, 3], utils.quat2euler(quat_from_rotation(R)) if np.all(np.isfinite(R)) else (np.nan,) * 3]
    cfg = job_config(name)
    work = tempfile.mkdtemp(prefix='uvs_sweep_')
    cwd = os.getcwd()
    t0 = time.time()
    try:
        os.makedirs(os.path.join(work, 'results', 'data'))
        with open(os.path.join(work, 'config.json'), 'w', encoding='utf-8') as fh:
            json.dump(cfg, fh)
        os.chdir(work)
        G.U.UR10Simulation = SweepPlant                                 # main.py:3 binds the name at import
        G._Clock.t = 0.0
        runpy.run_path(os.path.join(G.REF, 'main.py'), run_name='__main__')
        (path,) = glob.glob(os.path.join(work, 'results', 'data', '*', 'results.csv'))
        csv_bytes = os.path.getsize(path)
        rows, cell = reduce_csv(path)
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    seconds = time.time() - t0
    prov = {'note': 'per-trial / per-cell reduction of the results.csv the UNMODIFIED reference main.py wrote for this config '
                    '(oracle/gen_golden_sweep.py); no reference text', 'job': name, 'csv_bytes': csv_bytes,
            'reference_seconds': round(seconds, 1), 'numpy': np.__version__}
    np.savez_compressed(os.path.join(OUT, f'sweep_{name}.npz'), config=json.dumps(dict(cfg, _provenance=prov), sort_keys=True),
                        **rows, **{'cell_' + k: v for k, v in cell.items()})
    print(f'{name}: {len(rows["status"])} trials, {int(rows["status"].sum())} FAIL, {csv_bytes / 1e6:.0f} MB csv, {seconds:.0f} s', flush=True)


def main():
    names = sys.argv[1:] or list(JOBS)
    if len(names) == 1:
        return run_job(names[0])
    # MCKF jobs are the long ones: start them first; four reference processes at a time (8 cores, ~1 GB each while reducing)
    names.sort(key=lambda n: (JOBS[n][0] != 'MCKF', -JOBS[n][4]))
    running, env = [], dict(os.environ, OPENBLAS_NUM_THREADS='1', OMP_NUM_THREADS='1')
    while names or running:
        while names and len(running) < 4:
            running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), names.pop(0)], env=env))
        time.sleep(2)
        for p in list(running):
            if p.poll() is not None:
                running.remove(p)
                if p.returncode:
                    raise SystemExit(f'job failed with code {p.returncode}')


if __name__ == '__main__':
    main()

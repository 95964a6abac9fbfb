import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import uvs_amd as uvs
from oracle import c_oracle
import bench
import test_gpu_mckf_fpi as M
desired = bench.config2()['experiments']['desired_f']
T, K = 150, 90
q0, noise = M._mixed_batch(np.random.default_rng(77), T, K)
thr, cap = 0.1, 1000
ref = c_oracle.closed_loop_batch(q0, noise, desired, method='MCKF', steps=K, want_x=True, fpi_threshold=thr, fpi_epoch_max=cap)
for lanes in (0, 4, -2):
    fp = uvs.engine.make_params(8, 6, 'MCKF', 10.0, False, 0.05, 15.0, 0.2, desired, True, lanes, K, thr, cap)
    out = uvs.engine.closed_loop(fp, uvs.SyntheticPlant.ur10(desired).to_struct(), M._cuda(q0), M._cuda(noise.transpose(1, 2, 0)), want=('x', 'err', 'q'))
    X = out['x'].cpu().numpy().transpose(2, 0, 1); E = out['err'].cpu().numpy().transpose(2,0,1)
    for t in range(T):
        kd = int(ref['k_done'][t])
        if not kd: continue
        d = np.abs(X[t, :kd] - ref['X'][t, :kd]).max(axis=1) / np.abs(ref['X'][t, :kd]).max()
        if d.max() > 1e-8:
            k0 = int(np.argmax(d > 1e-8))
            print('lanes', lanes, 'trial', t, 'kd', kd, 'first bad step', k0, 'dev there', d[k0], 'max', d.max())
            print('  oracle fpi around', ref['fpi'][t, max(0,k0-2):k0+3], 'noise absmax at step', np.abs(noise[t, k0]).max(), np.abs(noise[t,k0-1]).max())
            print('  X ref row', ref['X'][t, k0, :6], '\n  X gpu row', X[t, k0, :6])
            dd = np.abs(X[t, k0] - ref['X'][t, k0]).reshape(8,6).max(axis=1); print('  per-row dev', dd)
            print('  err dev', np.abs(E[t,:kd]-ref['err'][t,:kd]).max(axis=1)[max(0,k0-1):k0+3])

import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import load_golden, scene_desired
import uvs_amd as uvs
for name in sys.argv[1:]:
    g = load_golden(name)
    meta, p = g['meta'], g['meta']['params']
    def fp(l): return uvs.engine.make_params(8, 6, 'MCKF', p['kernel_bw'], p['annealing'], meta['dt'], meta['t_max'], meta['gain'], g['desired'], True, l, None, p['fpi_threshold'], p['fpi_epoch_max'])
    plant = uvs.SyntheticPlant.ur10(scene_desired(g)).to_struct()
    T = 70
    q0 = torch.as_tensor(np.tile(g['q_start'], (T, 1)), device='cuda'); nz = torch.as_tensor(np.ascontiguousarray(np.repeat(g['noise_full'][:, :, None], T, axis=2)), device='cuda')
    for l in (2, 0):
        a = uvs.engine.closed_loop(fp(l), plant, q0, nz, want=('x', 'err'))
        print(name, 'lanes', l, 'status (2 = marked for the careful pass)', a['status'][:6].tolist(), 'k_done', a['k_done'][:3].tolist())

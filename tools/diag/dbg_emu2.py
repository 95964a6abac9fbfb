import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import load_golden, scene_desired
import uvs_amd as uvs
name = sys.argv[1] if len(sys.argv) > 1 else 'fpi_mckf_a1p0_bw1_fail'
g = load_golden(name)
meta, p = g['meta'], g['meta']['params']
METH = sys.argv[2] if len(sys.argv) > 2 else 'MCKF'
def fp(l): return uvs.engine.make_params(8, 6, METH, p['kernel_bw'], p['annealing'], meta['dt'], meta['t_max'], meta['gain'], g['desired'], True, l, None, p['fpi_threshold'], p['fpi_epoch_max'])
plant = uvs.SyntheticPlant.ur10(scene_desired(g)).to_struct()
T = 70
q0 = torch.as_tensor(np.tile(g['q_start'], (T, 1)), device='cuda'); nz = torch.as_tensor(np.ascontiguousarray(np.repeat(g['noise_full'][:, :, None], T, axis=2)), device='cuda')
a = uvs.engine.closed_loop(fp(2), plant, q0, nz, want=('x', 'err', 'q', 'f', 'dq'))
b = uvs.engine.closed_loop(fp(0), plant, q0, nz, want=('x', 'err', 'q', 'f', 'dq'))
kd = min(int(a['k_done'][0]), 3); print(g['q_start'], 'k_done', kd, int(b['k_done'][0]), 'status', int(a['status'][0]))
for key in ('x', 'err', 'q', 'f', 'dq'):
    A, B = a[key][:kd].cpu().numpy(), b[key][:kd].cpu().numpy()
    d = np.argwhere(A.view(np.int64) != B.view(np.int64))
    print(key, len(d), d[:6].tolist())
    for i in d[:6]:
        print('   ', repr(A[tuple(i)]), repr(B[tuple(i)]))
    if len(d):
        k0 = d[0][0]
        print('  first step', k0, 'components differing at that step (trial 0):', sorted(set(int(c) for kk, c, t in d if kk == k0 and t == 0)))
        print('  trials differing at that step:', sorted(set(int(t) for kk, c, t in d if kk == k0))[:20])

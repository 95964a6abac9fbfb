import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import uvs_amd
from conftest import load_golden, rel_err
np.set_printoptions(linewidth=200, precision=6)
name = 'rankdef_gmckf_dup_col'
g = load_golden(name)
meta, p = g['meta'], g['meta']['params']
K = len(g['t'])
f_seq = np.vstack([g['f_init'][None], g['f']])
cu = lambda a: torch.as_tensor(np.ascontiguousarray(a), device='cuda')
for T in (1, 3, 5, 40):
    fp = uvs_amd.engine.make_params(8, 6, meta['method'], p['kernel_bw'], p['annealing'], meta['dt'], meta['t_max'], meta['gain'], g['desired'], False, -2)
    out = uvs_amd.engine.replay(fp, cu(np.repeat(f_seq[:, :, None], T, axis=2)), cu(np.repeat(g['dq_prev'][:, :, None], T, axis=2)), cu(np.tile(g['X'][0], (T, 1))))
    cmd = out['dqcmd'].cpu().numpy()
    ref = g['dq_prev'][1:]
    print('T', T)
    for k in (6, 7, 8, 30, 43, 44, 45, 60):
        print('  k', k, 'ref', ref[k], '\n      t0 ', cmd[k, :, 0], '\n      tl ', cmd[k, :, T - 1], ' nan?', np.isnan(cmd[k]).any())
    s = np.array([np.linalg.svd(x.reshape(8, 6), compute_uv=False) for x in g['X'][5:10]])
    print((s[:, -1] / s[:, 0]))

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uvs_amd
from uvs_amd import engine, batch
import bench
T, dev = 65536, torch.device('cuda')
K = 299
for alpha in (1.0, 1.2, 1.5):
    cfg = bench.config2(); cfg['experiments']['epoch'] = T; cfg['noise']['noise_params']['alpha'] = alpha
    plan = batch.plan_trials(cfg, cells=[alpha])
    noise = batch.device_noise(cfg, plan, 0, T, K, dev)
    q0 = torch.as_tensor(plan.q_start.copy(), device=dev)
    plant = uvs_amd.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    for meth in ('GMCKF', 'KF', 'IMCCKF', 'MCKF'):
        fp = engine.make_params(8, 6, meth, 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0)
        out = engine.closed_loop(fp, plant, q0, noise, want=('err',))
        st = out['status'].cpu().numpy()
        kd = out['k_done'].cpu().numpy()
        sus = np.nonzero(st == 2)[0]
        waves16 = len(set(sus // 16))
        e = out['err'].abs().amax(dim=(0, 1)).cpu().numpy()
        print(f'alpha {alpha} {meth:7s}: suspect {len(sus):6d} ({100*len(sus)/T:.2f} %), fail {int((st==1).sum()):6d}; careful wavefronts (16 trials) with work: {waves16} of {T//16}; '
              f'max|err| median over suspects {np.median(e[sus]) if len(sus) else 0:.3g}, over the rest {np.median(e[st==0]):.3g}')

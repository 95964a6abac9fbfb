import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import uvs_amd
cfg = json.load(open('tests/golden/config_reference.json')); cfg['estimator']['method'] = 'MCKF'
for cells in ([1.0], [1.1818], [2.0]):
    res = uvs_amd.batch.run_batch(cfg, cells=cells, epoch=8192, want=('err',))
    st = res.stats.cpu().numpy()
    print('alpha', cells[0], 'wave-steps entering the branch: mean %.1f of 299' % st[::32, 0].mean(), ' trial-steps needing it: %.2f per trial' % st[:, 1].mean(), ' passes per wave %.1f' % st[::32, 2].mean(), 'max passes', st[:, 2].max())

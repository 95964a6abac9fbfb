import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import uvs_amd
cfg = json.load(open('tests/golden/config_reference.json'))
for method in ('MCKF', 'GMCKF'):
    cfg['estimator']['method'] = method
    uvs_amd.batch.run_batch(cfg, epoch=1, want=('err',))
    for cells in ([1.0], [1.0909], [1.1818], [1.5], [2.0]):
        res = uvs_amd.batch.run_batch(cfg, cells=cells, epoch=65536, want=('err',))
        res = uvs_amd.batch.run_batch(cfg, cells=cells, epoch=65536, want=('err',))
        print(method, 'alpha', cells[0], 'ms %.3f' % (res.seconds * 1e3), 'fail', int(res.status.sum()))

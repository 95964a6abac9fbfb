import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import uvs_amd
cfg = json.load(open('tests/golden/config_reference.json')); cfg['estimator']['method'] = 'MCKF'
uvs_amd.batch.run_batch(cfg, epoch=1, want=('err',))
for cells in ([1.0], [1.0909], [1.1818], [1.2727], [1.5], [2.0]):
    for lanes in (0, 4):
        res = uvs_amd.batch.run_batch(cfg, cells=cells, epoch=5461, want=('err',), lanes=lanes)
        res = uvs_amd.batch.run_batch(cfg, cells=cells, epoch=5461, want=('err',), lanes=lanes)
        print('alpha', cells[0], 'lanes', lanes, 'ms %.3f' % (res.seconds * 1e3), 'fail', int(res.status.sum()))

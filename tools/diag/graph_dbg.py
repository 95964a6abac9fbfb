import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import ctypes as C, numpy as np, torch
import uvs_amd as uvs
from conftest import load_golden
K, T = 100, 96
g = load_golden('closed_gmckf_a1p5')
rng = np.random.default_rng(21)
q0 = np.tile(g['q_start'], (T, 1)); q0[:, :3] += rng.uniform(-0.2, 0.1, (T, 3))
noise = rng.standard_t(1.5, size=(K, 8, T)) * rng.choice([0.3, 1.0, 3.0], size=T)
q0, noise = torch.as_tensor(q0, device='cuda'), torch.as_tensor(noise, device='cuda')
plant = uvs.SyntheticPlant.ur10(g['desired']).to_struct()
fp = uvs.engine.make_params(8, 6, 'MCKF', 10.0, True, 0.05, 15.0, 0.2, g['desired'], True, 2, K)
fp.reserved = 4 << 8
eager = uvs.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
torch.cuda.synchronize()
x = uvs.engine.alloc_stream(T, K, 48); err = uvs.engine.alloc_stream(T, K, 8); q = uvs.engine.alloc_stream(T, K, 6)
stats = torch.zeros((T, 3), dtype=torch.float64, device='cuda')
status = torch.zeros(T, dtype=torch.int32, device='cuda'); k_done = torch.zeros(T, dtype=torch.int32, device='cuda')
sv, NV, View = uvs.engine.stream_view, uvs.engine.NULL_VIEW, uvs.engine.View
flat = lambda t: View(t.data_ptr(), t.stride(0), 0, t.stride(1))
mode = sys.argv[1] if len(sys.argv) > 1 else 'graph'
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    uvs.engine.workspace(fp, plant, T, q0.device)
def launch():
    return uvs.engine.launch_closed_loop(fp, plant, T, flat(q0), sv(noise), NV, sv(x), sv(err), sv(q), NV, NV, stats.data_ptr(), status.data_ptr(), k_done.data_ptr(), NV, NV, device=q0.device)
if mode == 'graph':
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = launch()
    run = graph.replay
else:
    run = launch
for rep in range(4):
    for t in (x, err, q, stats): t.fill_(float('nan'))
    status.fill_(7); k_done.fill_(-1)
    run(); torch.cuda.synchronize()
    ok = (status == 0)
    for key, t in (('x', x), ('err', err), ('q', q)):
        a, b = t[:, :, ok], eager[key][:, :, ok]
        bad = (a.view(torch.int64) != b.contiguous().view(torch.int64)) if False else (a.contiguous().view(torch.int64) != b.contiguous().view(torch.int64))
        if bad.any():
            idx = bad.nonzero()
            print(mode, 'rep', rep, key, 'mismatches', int(bad.sum()), 'nan', int(torch.isnan(a[bad]).sum()), 'steps', sorted(set(idx[:, 0].tolist()))[:20], 'trials', sorted(set(idx[:, 2].tolist()))[:20], 'comps', sorted(set(idx[:, 1].tolist()))[:20])
            print('   a', a[bad][:4].tolist(), 'b', b[bad][:4].tolist())
        else:
            print(mode, 'rep', rep, key, 'ok')
    print('status eq', bool(torch.equal(status, eager['status'])), 'failed', int((status != 0).sum()))

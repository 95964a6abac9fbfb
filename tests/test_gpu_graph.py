"""The C ABI promises "nothing is allocated, freed or synchronised inside; work is enqueued on the stream passed in" (include/uvs_rmckf.h):
every entry point can therefore be captured into a HIP graph and replayed.  These tests hold the library to it -- the closed loop (with the
memset + kernel of a segmented MCKF launch, and with the careful second pass of rank-deficient trials), the replay, the statistics kernel
and a loop of per-step calls (experiment.py:166-312 once per time step: the launch-bound case a graph is for) are captured on a side
stream, replayed on fresh inputs, and must return the bits of the eager calls."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden, scene_desired

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def uvs():
    import torch
    assert torch.cuda.is_available()
    import uvs_amd
    return uvs_amd


def _bits(t):
    import torch
    return t.contiguous().view(torch.int64)


def _inputs(uvs, T, K, seed):
    import torch
    g = load_golden('closed_gmckf_a1p5')
    rng = np.random.default_rng(seed)
    q0 = np.tile(g['q_start'], (T, 1))
    q0[:, :3] += rng.uniform(-0.2, 0.1, (T, 3))
    noise = rng.standard_t(1.5, size=(K, 8, T)) * rng.choice([0.3, 1.0, 3.0], size=T)
    return g, torch.as_tensor(q0, device='cuda'), torch.as_tensor(noise, device='cuda')


@pytest.mark.parametrize('method,segments,options', [('GMCKF', 0, 0), ('MCKF', 4, 0), ('GMCKF', 4, 0), ('KF', 0, 0), ('GMCKF', 0, 1)])
def test_closed_loop_is_graph_capturable(uvs, method, segments, options):
    """options 1 = UVS_OPT_STRICT_PINV (every trial marked, then the careful kernels); segments: flags reset + segmented kernel."""
    import torch
    K, T = 100, 96
    g, q0, noise = _inputs(uvs, T, K, 21)
    plant = uvs.SyntheticPlant.ur10(scene_desired(g)).to_struct()
    fp = uvs.engine.make_params(8, 6, method, 10.0, True, 0.05, 15.0, 0.2, g['desired'], True, 2 if segments else 0, K)
    fp.reserved = (segments << 8) | options
    if segments:
        assert int(uvs.lib().uvs_rmckf_closed_loop_segments(C.byref(fp), C.byref(plant), T)) == segments
    eager = uvs.engine.closed_loop(fp, plant, q0, noise, want=('x', 'err', 'q'))
    torch.cuda.synchronize()

    # static buffers of the graph
    q_in, nz_in = torch.empty_like(q0), torch.empty_like(noise)
    x = uvs.engine.alloc_stream(T, K, 48); err = uvs.engine.alloc_stream(T, K, 8); q = uvs.engine.alloc_stream(T, K, 6)
    stats = torch.zeros((T, 3), dtype=torch.float64, device='cuda')
    status = torch.zeros(T, dtype=torch.int32, device='cuda'); k_done = torch.zeros(T, dtype=torch.int32, device='cuda')
    sv, NV, View = uvs.engine.stream_view, uvs.engine.NULL_VIEW, uvs.engine.View
    flat = lambda t: View(t.data_ptr(), t.stride(0), 0, t.stride(1))      # noqa: E731
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        uvs.engine.workspace(fp, plant, T, q0.device)                      # the caller's workspace exists before the capture starts
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = uvs.engine.launch_closed_loop(fp, plant, T, flat(q_in), sv(nz_in), NV, sv(x), sv(err), sv(q), NV, NV,
                                           stats.data_ptr(), status.data_ptr(), k_done.data_ptr(), NV, NV, device=q0.device)
    assert rc == 0, uvs.lib().uvs_last_error()
    for rep in range(2):                                                    # replay twice: flags / marks of the first run must not leak
        q_in.copy_(q0); nz_in.copy_(noise)
        for t in (x, err, q, stats):
            t.fill_(float('nan'))
        status.fill_(7); k_done.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(status, eager['status']) and torch.equal(k_done, eager['k_done']), rep
        ok = status == 0
        assert torch.equal(_bits(stats[ok]), _bits(eager['stats'][ok]))
        for key, t in (('x', x), ('err', err), ('q', q)):
            assert torch.equal(_bits(t[:, :, ok]), _bits(eager[key][:, :, ok])), (rep, key)
    # new inputs through the same graph
    g2, q0b, noiseb = _inputs(uvs, T, K, 22)
    eager_b = uvs.engine.closed_loop(fp, plant, q0b, noiseb, want=('x',))
    q_in.copy_(q0b); nz_in.copy_(noiseb)
    graph.replay()
    torch.cuda.synchronize()
    ok = eager_b['status'] == 0
    assert torch.equal(status, eager_b['status']) and torch.equal(_bits(x[:, :, ok]), _bits(eager_b['x'][:, :, ok]))


def test_careful_second_pass_inside_a_graph(uvs):
    """Rank-deficient X0 (the reference's rank-4 product fixture): the main kernel marks the trial, the careful kernels redo it with numpy's
    pinv cutoff (experiment.py:312) -- both passes are enqueued unconditionally, so the capture holds the pair and the replay matches the
    eager run and the reference's trajectory."""
    import torch
    g, h = load_golden('rankdef_gmckf_rank4_product'), load_golden('closed_gmckf_a1p5')
    K, T = 120, 40
    sick = [0, 21]
    plant = uvs.SyntheticPlant.ur10(scene_desired(g)).to_struct()
    meta, p = g['meta'], g['meta']['params']
    fp = uvs.engine.make_params(8, 6, meta['method'], p['kernel_bw'], p['annealing'], meta['dt'], meta['t_max'], meta['gain'], g['desired'], False, 0, K)
    rng = np.random.default_rng(4)
    x0 = np.tile(h['X'][0], (T, 1)) * (1 + 0.02 * rng.standard_normal((T, 1)))
    noise = rng.standard_t(3, size=(K, 8, T))
    for t in sick:
        x0[t] = g['X'][0]
        noise[:, :, t] = g['noise'][:K]
    q0 = torch.as_tensor(np.tile(g['q_start'], (T, 1)), device='cuda')
    x0, noise = torch.as_tensor(x0, device='cuda'), torch.as_tensor(noise, device='cuda')
    eager = uvs.engine.closed_loop(fp, plant, q0, noise, x0=x0, want=('err', 'dq'))
    torch.cuda.synchronize()
    err = uvs.engine.alloc_stream(T, K, 8); dq = uvs.engine.alloc_stream(T, K, 6)
    stats = torch.zeros((T, 3), dtype=torch.float64, device='cuda')
    status = torch.zeros(T, dtype=torch.int32, device='cuda'); k_done = torch.zeros(T, dtype=torch.int32, device='cuda')
    sv, NV, View = uvs.engine.stream_view, uvs.engine.NULL_VIEW, uvs.engine.View
    flat = lambda t: View(t.data_ptr(), t.stride(0), 0, t.stride(1))      # noqa: E731
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = uvs.engine.launch_closed_loop(fp, plant, T, flat(q0), sv(noise), flat(x0), NV, sv(err), NV, NV, sv(dq),
                                           stats.data_ptr(), status.data_ptr(), k_done.data_ptr(), NV, NV, device=q0.device)
    assert rc == 0, uvs.lib().uvs_last_error()
    for rep in range(2):
        err.fill_(float('nan')); dq.fill_(float('nan')); status.fill_(3)
        graph.replay()
        torch.cuda.synchronize()
        assert not status.any() and torch.equal(k_done, eager['k_done'])
        for key, t in (('err', err), ('dq', dq)):
            assert torch.equal(_bits(t), _bits(eager[key])), (rep, key)
        assert torch.equal(_bits(stats), _bits(eager['stats']))
    for t in sick:                                                          # numpy's truncated command, through the graph
        ref = np.asarray(g['err'])[:K]
        assert np.abs(err[:, :, t].cpu().numpy() - ref).max() <= 1e-7 * np.abs(ref).max()


def test_step_loop_as_one_graph(uvs):
    """experiment.py:166-312 once per time step with the state in HBM (uvs_rmckf_step_f64): 40 steps over recorded features as ONE graph
    launch -- what a caller with a launch-bound loop does -- against 40 eager calls."""
    import torch
    g = load_golden('closed_gmckf_a1p5')
    T, K = 17, 40
    f_rec = torch.as_tensor(np.asarray(g['f'])[:K + 1], device='cuda')                                      # [step][m]
    rng = np.random.default_rng(8)
    f_all = (f_rec[:, None, :] + torch.as_tensor(rng.normal(size=(K + 1, T, 8)) * 0.5, device='cuda')).contiguous()
    dq_all = torch.as_tensor(rng.normal(size=(K, T, 6)) * 0.01, device='cuda')
    x0 = rng.normal(size=(T, 48)) * 30.0

    def fp_():
        return uvs.engine.make_params(8, 6, 'GMCKF', 10.0, True, 0.05, 15.0, 0.2, g['desired'], False, 0)

    eager = uvs.engine.FilterBank(fp_(), T, x0)
    eager_dq = []
    for k in range(K):
        dq, err, kappa, status = eager.step(f_all[k + 1], f_all[k], dq_all[k], k)
        eager_dq.append(dq.clone())
    torch.cuda.synchronize()

    bank = uvs.engine.FilterBank(fp_(), T, x0)
    out = torch.zeros((K, T, 6), dtype=torch.float64, device='cuda')
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for k in range(K):
            dq, err, kappa, status = bank.step(f_all[k + 1], f_all[k], dq_all[k], k)
            out[k].copy_(dq)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(_bits(out), _bits(torch.stack(eager_dq)))
    assert torch.equal(_bits(bank.X), _bits(eager.X)) and torch.equal(_bits(bank.P), _bits(eager.P))
    assert torch.equal(bank.status, eager.status)


def test_replay_and_statistics_are_graph_capturable(uvs):
    import torch
    g = load_golden('closed_gmckf_a1p5')
    T, K = 64, 60
    rng = np.random.default_rng(10)
    f = torch.as_tensor(rng.normal(size=(K + 1, 8, T)) * 20 + 300, device='cuda')
    dq = torch.as_tensor(rng.normal(size=(K, 6, T)) * 0.02, device='cuda')
    x0 = torch.as_tensor(rng.normal(size=(T, 48)) * 30.0, device='cuda')
    fp = uvs.engine.make_params(8, 6, 'GMCKF', 10.0, False, 0.05, 15.0, 0.2, g['desired'], False, 0, K)
    eager = uvs.engine.replay(fp, f, dq, x0, want=('x', 'err', 'dqcmd'))
    t = uvs.engine.loop_clock(0.05, 15.0)[:K]
    eager_stats = uvs.engine.stats_reduce(eager['err'], t)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                            # warm the allocator of the side stream; t uploaded once
        t_dev = torch.as_tensor(np.asarray(t, float), device='cuda')
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        cap = uvs.engine.replay(fp, f, dq, x0, want=('x', 'err', 'dqcmd'))
        stats = torch.empty((T, 3), dtype=torch.float64, device='cuda')
        rc = uvs.lib().uvs_stats_reduce_f64(T, K, 8, uvs.engine.stream_view(cap['err']), t_dev.data_ptr(), None, stats.data_ptr(),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, uvs.lib().uvs_last_error()
    graph.replay()
    torch.cuda.synchronize()
    for key in ('x', 'err', 'dqcmd', 'status', 'k_done'):
        a, b = cap[key], eager[key]
        assert torch.equal(a, b) if a.dtype == torch.int32 else torch.equal(_bits(a), _bits(b)), key
    assert torch.equal(_bits(stats), _bits(eager_stats))


def test_sweep_cell_with_shared_noise_is_one_graph(uvs):
    """A whole sweep cell -- device seeding (uvs_pcg64_seed_u64), the T + 70 shared noise streams in chunks (uvs_noise_generate_streams_f64, round 5)
    and the closed loop reading them through the overlapping view -- captured as ONE graph; replays return the bits of the eager pipeline, and with
    another seed vector in the static input buffer those of the eager pipeline on those seeds."""
    import torch
    import bench
    nd = uvs.noise_device
    cfg = bench.config2()
    T, K, M = 333, 80, 8
    des = cfg['experiments']['desired_f']
    plant = uvs.SyntheticPlant.ur10(des).to_struct()
    fp = uvs.engine.make_params(8, 6, 'GMCKF', 10.0, False, 0.05, 15.0, 0.2, des, True, 2, K)
    plan = uvs.batch.plan_trials(cfg, cells=[1.5], epoch=T)
    q0 = torch.as_tensor(plan.q_start.copy(), device='cuda')
    params = dict(alpha=1.5, beta=0, gamma=1, delta=0)
    S = T + nd.SEED_STEP * (M - 1)
    q = nd.make_noise_params(uvs.NoiseType.ALPHA_STABLE, params, M, K)
    zig = nd._zig('cuda')

    def eager(seed0):
        _, view = nd.generate_shared(uvs.NoiseType.ALPHA_STABLE, params, seed0, T, M, K)
        return uvs.engine.closed_loop(fp, plant, q0, view, want=('x', 'err'))

    ref_a, ref_b = eager(123456), eager(777)
    torch.cuda.synchronize()
    seeds = torch.empty(S, dtype=torch.int64, device='cuda')
    states = torch.empty((S, 4), dtype=torch.int64, device='cuda')
    buf = torch.empty((K, S), dtype=torch.float64, device='cuda')
    view = torch.as_strided(buf, (K, M, T), (S, nd.SEED_STEP, 1))
    x = uvs.engine.alloc_stream(T, K, 48); err = uvs.engine.alloc_stream(T, K, 8)
    stats = torch.zeros((T, 3), dtype=torch.float64, device='cuda')
    status = torch.zeros(T, dtype=torch.int32, device='cuda'); k_done = torch.zeros(T, dtype=torch.int32, device='cuda')
    sv, NV, View = uvs.engine.stream_view, uvs.engine.NULL_VIEW, uvs.engine.View
    flat = lambda t: View(t.data_ptr(), t.stride(0), 0, t.stride(1))      # noqa: E731
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc1 = uvs.lib().uvs_pcg64_seed_u64(S, seeds.data_ptr(), states.data_ptr(), st)
        rc2 = uvs.lib().uvs_noise_generate_streams_f64(C.byref(q), S, states.data_ptr(), zig.data_ptr(), buf.data_ptr(), 1, S, st)
        rc3 = uvs.engine.launch_closed_loop(fp, plant, T, flat(q0), sv(view), NV, sv(x), sv(err), NV, NV, NV,
                                            stats.data_ptr(), status.data_ptr(), k_done.data_ptr(), NV, NV, device=q0.device)
    assert (rc1, rc2, rc3) == (0, 0, 0), uvs.lib().uvs_last_error()
    for seed0, ref in ((123456, ref_a), (777, ref_b), (123456, ref_a)):
        seeds.copy_(torch.arange(S, dtype=torch.int64, device='cuda') + seed0)
        for t in (x, err, stats, buf):
            t.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(status, ref['status']) and torch.equal(k_done, ref['k_done']) and int(status.sum()) == 0
        assert torch.equal(_bits(x), _bits(ref['x'])) and torch.equal(_bits(err), _bits(ref['err'])) and torch.equal(_bits(stats), _bits(ref['stats']))

"""On-device noise generator (csrc/noise_kernels.hpp) against the reference's NoiseProfiler streams (tests/golden/noise_*.npz) and
against the host generator, through uvs_noise_generate_f64."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

# streams built only from PCG64 doubles, the ziggurat normal and +,*: bit-identical to numpy (a tail sample may differ by an ulp)
EXACT = {'noise_white', 'noise_white_m2', 'noise_mixture', 'noise_mixture_hold', 'noise_bimodal', 'noise_bimodal_hold', 'noise_alpha2p0',
         'noise_uniform_jitter', 'noise_white_seed_plus10', 'noise_uniform_jitter_seed_plus10'}


@pytest.fixture(scope='module')
def uvs():
    import uvs_amd
    return uvs_amd


@pytest.mark.parametrize('layout', ['kct', 'tkc'])
@pytest.mark.parametrize('name', golden_names('noise_'))
def test_device_streams_match_reference(uvs, name, layout):
    g = load_golden(name)
    meta = g['meta']
    K = len(g['values'])
    seeds = [meta['seed'], meta['seed'] + 1, meta['seed'] + 10]
    out = uvs.noise_device.generate(uvs.NoiseType[meta['noise_type']], meta['noise_params'], seeds, meta['m'], K, meta['hold'], meta['hold_cnt'], layout)
    got = uvs.engine.as_tkc(out, layout).cpu().numpy()
    if name in EXACT:
        bad = got[0] != g['values']
        assert bad.mean() < 1e-3 and np.allclose(got[0], g['values'], rtol=4e-16, atol=0)
    else:
        assert np.allclose(got[0], g['values'], rtol=2e-13, atol=0)
    # seed aliasing of the reference: trial seed+10 feature i == trial seed feature i+1 (noise.py:70), when no hold couples the pairs
    if not meta['hold'] and meta['noise_type'] in ('WHITE_NOISE', 'ALPHA_STABLE', 'UNIFORM') and meta['m'] > 2:
        assert np.array_equal(got[2][:, 0], got[0][:, 1])
    assert not np.array_equal(got[0], got[1])


@pytest.mark.parametrize('name', [n for n in golden_names('noise_') if 'alpha' in n])
def test_as_written_variant_matches_reference_within_a_few_ulp(uvs, name):
    """UVS_NOISE_OPT_AS_WRITTEN (round 6): the Chambers-Mallows-Stuck powers evaluated as noise.py:188-199 writes them -- the reference's alpha-stable
    fixtures within 6 ulp sample by sample (the folded default: 2e-13 relative, i.e. hundreds of ulp allowed), the shared-stream generator
    included; special cases (alpha = 2, Cauchy) are untouched by the bit."""
    g = load_golden(name)
    meta = g['meta']
    K = len(g['values'])
    nt = uvs.NoiseType[meta['noise_type']]
    out = uvs.noise_device.generate(nt, meta['noise_params'], [meta['seed'], meta['seed'] + 1], meta['m'], K, meta['hold'], meta['hold_cnt'], as_written=True)
    got = uvs.engine.as_tkc(out).cpu().numpy()[0]
    ulp = np.abs(np.ascontiguousarray(got).view(np.int64) - np.ascontiguousarray(g['values']).view(np.int64))
    q = uvs.noise_device.make_noise_params(nt, meta['noise_params'], meta['m'], K, as_written=True)
    a, b = meta['noise_params'].get('alpha'), meta['noise_params'].get('beta', 0)
    general = a not in (1, 2) and not (a == 0.5 and abs(b) == 1)
    if general:
        # (gamma x + delta cancels near zero when delta != 0, noise.py:205: the 6 ulp are those of |value| + |delta|)
        room = np.spacing(np.abs(g['values']) + abs(meta['noise_params'].get('delta', 0.0)))
        worst = float((np.abs(got - g['values']) / room).max())
        print(f'{name}: as-written variant max {worst:.1f} ulp, exact {float((ulp == 0).mean()):.3f}')
        assert worst <= 6, (name, worst)
    else:                                                           # the bit changes nothing there: the default kernels' gate
        assert np.allclose(got, g['values'], rtol=2e-13, atol=0)
    assert q.type == 4 | 0x100 and uvs.lib().uvs_noise_kernel_variant(q) == (2 if general else 0)
    if not meta['hold'] and meta['m'] > 2:
        _, view = uvs.noise_device.generate_shared(nt, meta['noise_params'], meta['seed'], 2, meta['m'], K, as_written=True)
        assert np.array_equal(view.permute(2, 0, 1).cpu().numpy()[0], got)


@pytest.mark.parametrize('kind,params,hold', [
    ('ALPHA_STABLE', dict(alpha=1.5, beta=0, gamma=1, delta=0), False),
    ('ALPHA_STABLE', dict(alpha=1.0, beta=0, gamma=1, delta=0), True),
    ('ALPHA_STABLE', dict(alpha=1.7, beta=-0.3, gamma=0.5, delta=2.0), True),
    ('GAUSSIAN_MIXTURE', dict(std=1.0, mean=50.0, rho=0.1), True),
    ('GAUSSIAN_BIMODAL', dict(std=2.0, mean=30.0, rho=0.3), True),
    ('WHITE_NOISE', dict(std=3.0), False)])
def test_device_batch_matches_host_batch(uvs, kind, params, hold):
    nt = uvs.NoiseType[kind]
    seeds = 987654 + 7 * np.arange(300)
    K = 120
    host = uvs.noise_batch(nt, params, seeds, 8, K, hold, 10)
    dev = uvs.engine.as_tkc(uvs.noise_device.generate(nt, params, seeds, 8, K, hold, 10)).cpu().numpy()
    assert np.allclose(dev, host, rtol=2e-13, atol=1e-12 * (1 + abs(params.get('delta', 0.0))))     # gamma x + delta cancels near 0
    if kind in ('GAUSSIAN_MIXTURE', 'GAUSSIAN_BIMODAL', 'WHITE_NOISE'):
        assert (dev != host).mean() < 1e-3


def test_closed_loop_on_device_noise_matches_host_noise(uvs):
    """The Monte-Carlo driver with on-device streams reproduces the run on host-generated streams (non-chaotic config 2)."""
    import bench
    cfg = bench.config2()
    cfg['experiments']['epoch'] = 256
    a = uvs.batch.run_batch(cfg, cells=[1.5], want=('err',), noise_on_device=True)
    b = uvs.batch.run_batch(cfg, cells=[1.5], want=('err',), noise_on_device=False)
    ea, eb = a.streams['err'].cpu().numpy(), b.streams['err'].cpu().numpy()
    dev = np.abs(ea - eb).max(axis=(0, 1)) / np.abs(eb).max(axis=(0, 1))
    assert np.median(dev) < 1e-11 and dev.max() < 1e-6
    assert np.array_equal(a.status.cpu().numpy(), b.status.cpu().numpy())


def test_device_seeding_matches_numpy(uvs):
    """uvs_pcg64_seed_u64: SeedSequence + PCG64 seeding on the device against numpy's own generator states, including seeds
    with a second entropy word (>= 2**32) and the wrap of seed + 10 j at 2**64."""
    import torch
    NT = uvs.NoiseType
    rng = np.random.default_rng(3)
    seeds = np.concatenate([np.array([0, 1, 2, 123456, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 63, 2 ** 64 - 1], dtype=np.uint64),
                            rng.integers(0, 2 ** 63, 40, dtype=np.uint64), rng.integers(0, 2 ** 31, 40, dtype=np.uint64)])
    st = torch.empty((len(seeds), 4), dtype=torch.int64, device='cuda')
    sd = torch.as_tensor(seeds.view(np.int64), device='cuda')
    uvs._lib.check(uvs.lib().uvs_pcg64_seed_u64(len(seeds), sd.data_ptr(), st.data_ptr(), None))
    got = st.cpu().numpy().view(np.uint64)
    for s, row in zip(seeds, got):
        ref = np.random.PCG64(int(s)).state['state']
        assert (int(row[0]) << 64) | int(row[1]) == ref['state'] and (int(row[2]) << 64) | int(row[3]) == ref['inc'], int(s)
    assert np.array_equal(got, uvs.pcg.pcg64_states(seeds))
    # the generator table of a mixture trial: seed + 10 j for 3 m generators, then 2 seed + i for the m selectors (noise.py:59,70)
    trial_seeds = np.array([123456, 2 ** 64 - 7], dtype=np.uint64)
    dev = uvs.noise_device.device_generator_states(NT.GAUSSIAN_BIMODAL, trial_seeds, 8).cpu().numpy().view(np.uint64)
    with np.errstate(over='ignore'):
        host = uvs.pcg.pcg64_states(uvs.noise_device.generator_seeds(NT.GAUSSIAN_BIMODAL, trial_seeds, 8))
    assert dev.shape == (2, 32, 4) and np.array_equal(dev, host)


@pytest.mark.parametrize('alpha,beta', [(0.8, 0.0), (0.8, -0.7), (1.7, 0.3), (1.95, 0.0), (0.3, 1.0)])
def test_device_cms_matches_host_generator_outside_the_fixtures(uvs, alpha, beta):
    """Chambers-Mallows-Stuck branch (both powers folded into one exponential on the device) against the host NoiseProfiler (numpy,
    formula as written in noise.py:188-199) for index / skewness pairs the reference fixtures do not cover, incl. alpha < 1."""
    params = dict(alpha=alpha, beta=beta, gamma=1.3, delta=-0.4)
    seeds = [77, 78, 123456]
    K = 400
    dev = uvs.engine.as_tkc(uvs.noise_device.generate(uvs.NoiseType.ALPHA_STABLE, params, seeds, 8, K), 'kct').cpu().numpy()
    host = uvs.noise_batch(uvs.NoiseType.ALPHA_STABLE, params, seeds, 8, K)
    finite = np.isfinite(host)
    assert finite.mean() > 0.999 and np.array_equal(np.isfinite(dev), finite)
    scale = np.maximum(np.abs(host[finite]), 1e-3)                           # delta shifts values through zero: relative to max(|x|, 1e-3)
    assert np.max(np.abs(dev[finite] - host[finite]) / scale) <= 5e-13


@pytest.mark.parametrize('alpha', [0.05, 0.3, 0.75, 1.0909090909090908, 1.5, 1.9090909090909092, 1.99, 1.9999])
def test_symmetric_stable_fast_path_and_its_gate(uvs, alpha):
    """beta = 0: the specialised instantiation (own log / exp routines, cos((1 - alpha) V) by the addition theorem) where the launcher's
    error bound admits it (0.3 ... 1.99 here), the general kernel elsewhere (0.05, 1.9999) -- 400 k samples each against the host
    NoiseProfiler port at the gate of the reference fixtures."""
    params = dict(alpha=alpha, beta=0.0, gamma=1.0, delta=0.0)
    seeds = list(range(5000, 5128))
    K = 400
    dev = uvs.engine.as_tkc(uvs.noise_device.generate(uvs.NoiseType.ALPHA_STABLE, params, seeds, 8, K), 'kct').cpu().numpy()
    host = uvs.noise_batch(uvs.NoiseType.ALPHA_STABLE, params, seeds, 8, K)
    finite = np.isfinite(host)
    assert finite.mean() > 0.999 and np.array_equal(np.isfinite(dev), finite)
    rel = np.abs(dev[finite] - host[finite]) / np.abs(host[finite])
    assert rel.max() <= 2e-13, (alpha, rel.max())


# ---------------------------------------------------------------------------------------------- shared streams (uvs_noise_generate_streams_f64)
@pytest.mark.parametrize('kind,params', [
    ('ALPHA_STABLE', dict(alpha=1.5, beta=0, gamma=1, delta=0)), ('ALPHA_STABLE', dict(alpha=1.0, beta=0, gamma=1, delta=0)),
    ('ALPHA_STABLE', dict(alpha=2.0, beta=0, gamma=1, delta=0)), ('ALPHA_STABLE', dict(alpha=1.3, beta=0.4, gamma=2.0, delta=1.0)),
    ('WHITE_NOISE', dict(std=1.0)), ('UNIFORM', {})])
@pytest.mark.parametrize('T', [1, 70, 4099])
def test_shared_streams_are_the_per_trial_streams(uvs, kind, params, T):
    """T consecutive seeds, no hold: T + 70 streams generated once and read through the overlapping (S, 10, 1)-strided view carry the very
    bits of the 8 T streams generated per trial (noise.py:70 + main.py:137-139)."""
    import torch
    nt = uvs.NoiseType[kind]
    K, m, seed0 = 61, 8, 123456
    seeds = seed0 + np.arange(T)
    assert uvs.noise_device.shares_streams(nt, False, seeds) and not uvs.noise_device.shares_streams(nt, True, seeds)
    dense = uvs.noise_device.generate(nt, params, seeds, m, K)
    buf, view = uvs.noise_device.generate_shared(nt, params, seed0, T, m, K)
    assert buf.shape == (K, T + 70) and view.shape == dense.shape == (K, m, T)
    assert torch.equal(view.contiguous().view(torch.int64), dense.view(torch.int64))


def test_shared_streams_reproduce_the_reference_fixtures(uvs):
    g0, g1 = load_golden('noise_alpha1p5'), load_golden('noise_alpha1p5_seed_plus10')
    K = len(g0['values'])
    _, view = uvs.noise_device.generate_shared(uvs.NoiseType.ALPHA_STABLE, g0['meta']['noise_params'], g0['meta']['seed'], 11, 8, K)
    got = view.cpu().numpy()                                                  # [K][8][11]
    assert np.allclose(got[:, :, 0], g0['values'], rtol=2e-13, atol=0) and np.allclose(got[:, :, 10], g1['values'], rtol=2e-13, atol=0)


def test_streams_entry_point_rejects_what_does_not_alias(uvs):
    import ctypes as C
    import torch
    buf = torch.zeros((4, 80), dtype=torch.float64, device='cuda')
    st = torch.zeros((80, 4), dtype=torch.int64, device='cuda')
    zig = uvs.noise_device._zig('cuda')
    for kind, hold in (('GAUSSIAN_MIXTURE', 0), ('GAUSSIAN_BIMODAL', 0), ('ALPHA_STABLE', 10)):
        q = uvs.noise_device.make_noise_params(uvs.NoiseType[kind], dict(std=1.0, mean=5.0, rho=0.1, alpha=1.5, beta=0, gamma=1, delta=0), 8, 4, hold > 0, hold)
        assert uvs.lib().uvs_noise_generate_streams_f64(C.byref(q), 80, st.data_ptr(), zig.data_ptr(), buf.data_ptr(), 1, 80, None) == -1
    assert not uvs.noise_device.shares_streams(uvs.NoiseType.GAUSSIAN_MIXTURE, False, [5, 6, 7])
    assert not uvs.noise_device.shares_streams(uvs.NoiseType.ALPHA_STABLE, False, [5, 7, 8])          # seeds not consecutive


@pytest.mark.parametrize('T', [4099, 65536])
def test_sweep_rows_do_not_change_with_shared_streams(uvs, T):
    """The 12-cell sweep of main.py:104-148 with the noise of every cell generated once per distinct seed: per-trial [ISE, IAE, ITAE, status,
    k_done] rows and the error stream bit-identical to the sweep that generates all 8 T streams of a cell."""
    import torch
    import bench
    cfg = bench.config2()
    cfg['experiments']['epoch'] = T
    plan = uvs.batch.plan_trials(cfg)
    assert len(plan.cells) == 12
    K = len(uvs.engine.loop_clock(0.05, 15))
    fp = uvs.engine.make_params(8, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, cfg['experiments']['desired_f'], True, 0)
    plant = uvs.SyntheticPlant.ur10(cfg['experiments']['desired_f']).to_struct()
    failed = 0
    for c in range(12 if T < 10000 else 3):                                  # (the full-size run checks three cells: Cauchy, 1.09, 1.18)
        lo, hi = c * T, (c + 1) * T
        q0 = torch.as_tensor(plan.q_start[lo:hi].copy(), device='cuda')
        outs = []
        for share in (False, True):
            noise = uvs.batch.device_noise(cfg, plan, lo, hi, K, 'cuda', share=share)
            assert (noise.stride(2) == 1 and noise.stride(1) == 10) == share
            outs.append(uvs.engine.closed_loop(fp, plant, q0, noise, want=('err',)))
            del noise
        a, b = outs
        assert torch.equal(a['status'], b['status']) and torch.equal(a['k_done'], b['k_done'])
        assert torch.equal(a['stats'].view(torch.int64), b['stats'].view(torch.int64))
        assert torch.equal(a['err'].view(torch.int64), b['err'].view(torch.int64))
        failed += int((a['status'] != 0).sum())
        del outs, a, b
    assert failed < 0.01 * T

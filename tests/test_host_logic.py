"""CPU-only tests: the C-ABI library loads and exports every symbol of include/uvs_rmckf.h (no compute calls), and the host
logic around it (config validation, trial enumeration, loop clock, batched noise, statistics, API surface)."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_names, load_golden


@pytest.fixture(scope='module')
def uvs():
    import uvs_amd
    return uvs_amd


def test_library_exports_every_header_symbol(uvs):
    header = open(os.path.join(ROOT, 'include', 'uvs_rmckf.h')).read()
    declared = set(re.findall(r'\b(uvs_[a-z0-9_]+)\s*\(', header))
    assert declared == set(uvs._lib.SYMBOLS), 'ctypes table and header disagree'
    handle = uvs.lib()
    for name in declared:
        assert getattr(handle, name) is not None
    assert handle.uvs_version().startswith(b'uvs_rmckf')


def test_struct_layouts_match_the_header(uvs):
    # sizes implied by include/uvs_rmckf.h (LP64): catches a drifted ctypes mirror before it corrupts a launch
    assert ctypes.sizeof(uvs._lib.View) == 32
    assert ctypes.sizeof(uvs._lib.FilterParams) == 8 * 4 + 5 * 8 + 8 + 2 * 4 + 32 * 8
    assert ctypes.sizeof(uvs._lib.Plant) == 8 + 5 * 8 * 8 + 16 * 3 * 8 + 16 + 8 + 24


def test_argument_errors_are_reported_not_thrown(uvs):
    lib = uvs.lib()
    V = uvs._lib.NULL_VIEW
    rc = lib.uvs_rmckf_replay_f64(None, 4, V, V, V, V, V, V, V, None, None, V, V, None)
    assert rc == -1 and b'NULL' in lib.uvs_last_error()
    fp = uvs.engine.make_params(5, 3, 'GMCKF', desired=np.zeros(5), steps=3)          # shape that is not instantiated
    rc = lib.uvs_rmckf_replay_f64(ctypes.byref(fp), 4, V, V, V, V, V, V, V, None, None, V, V, None)
    assert rc == -2
    fp = uvs.engine.make_params(8, 6, 'GMCKF', desired=np.zeros(8), steps=3)
    fp.method = 1                                                                      # ANALYTICAL is not an estimator
    rc = lib.uvs_rmckf_replay_f64(ctypes.byref(fp), 4, V, V, V, V, V, V, V, None, None, V, V, None)
    assert rc == -4
    assert lib.uvs_supported_lanes(8, 6, None, 0) >= 4 and lib.uvs_supported_lanes(7, 7, None, 0) == 0


def test_workspace_query_is_host_logic(uvs):
    """uvs_rmckf_closed_loop_workspace_bytes: which launches the library would cut into segments (MCKF on the tuned two-lane kernel, DH plant,
    more than one round of wavefronts) and how much caller-owned scratch that takes -- no GPU involved."""
    lib = uvs.lib()
    plant = uvs.SyntheticPlant.ur10().to_struct()
    q = lambda fp, T: int(lib.uvs_rmckf_closed_loop_workspace_bytes(ctypes.byref(fp), ctypes.byref(plant), T))   # noqa: E731
    per_chunk = (3 * 6 + 6 + 4 + 1 + 4 * 21 + 4 * 6 + 12 + 1) * 64 * 8
    seg = lambda fp, T: int(lib.uvs_rmckf_closed_loop_segments(ctypes.byref(fp), ctypes.byref(plant), T))      # noqa: E731
    mckf = uvs.engine.make_params(8, 6, 'MCKF', desired=np.zeros(8))
    assert q(mckf, 32768) == 0                                             # one round of wavefronts: nothing to balance
    assert [seg(mckf, T) for T in (32768, 32769, 65536, 98304, 98305, 1048576)] == [1, 8, 8, 8, 4, 4]
    flags = lambda chunks: ((chunks + 1) * 4 + 255) // 256 * 256          # noqa: E731 -- one hand-over counter per chunk + the fallback count (round 6)
    assert q(mckf, 65536) == flags(2048) + 2048 * per_chunk                # flags + state
    assert q(mckf, 65537) == flags(2049) + 2049 * per_chunk
    off = lambda fp, T: int(lib.uvs_rmckf_closed_loop_fallback_offset(ctypes.byref(fp), ctypes.byref(plant), T))   # noqa: E731
    assert off(mckf, 65536) == 2048 * 4 and off(mckf, 32768) == 0 and off(mckf, 65537) == 2049 * 4               # right behind the chunks' counters; 0 = not segmented
    assert q(uvs.engine.make_params(8, 6, 'GMCKF', desired=np.zeros(8)), 65536) == 0 and q(uvs.engine.make_params(8, 6, 'KF', desired=np.zeros(8)), 65536) == 0
    # RMCKF: wavefronts of equal length, so only launches that are not a whole number of rounds (1 024 wavefronts of 32 trials) are cut -- in four
    rm = uvs.engine.make_params(8, 6, 'GMCKF', desired=np.zeros(8))
    assert [seg(rm, T) for T in (16384, 32768, 36000, 40000, 49152, 65536, 65536 + 2048, 81920, 98304, 131072 + 16384, 196608 + 16384, 16385)] == [1, 1, 1, 4, 4, 1, 1, 4, 1, 4, 1, 1]
    assert q(rm, 49152) == flags(1536) + 1536 * per_chunk and seg(uvs.engine.make_params(8, 6, 'KF', desired=np.zeros(8)), 49152) == 1
    rm.reserved = 1                                                        # strict pinv: everything goes to the careful kernels, nothing to cut
    assert seg(rm, 49152) == 1
    assert q(uvs.engine.make_params(8, 6, 'MCKF', desired=np.zeros(8), lanes=4), 65536) == 0 and q(uvs.engine.make_params(8, 6, 'MCKF', desired=np.zeros(8), lanes=-2), 65536) == 0
    forced = uvs.engine.make_params(8, 6, 'MCKF', desired=np.zeros(8)); forced.reserved = 3 << 8
    assert q(forced, 40) == 256 + 2 * per_chunk
    forced.reserved = 1 << 8
    assert q(forced, 65536) == 0
    short = uvs.engine.make_params(8, 6, 'MCKF', desired=np.zeros(8), steps=20); short.reserved = 4 << 8
    assert q(short, 65536) == 0                                            # nothing to cut in a 20-step trial
    linp = uvs._lib.Plant(); linp.kind = uvs._lib.PLANT_LINEAR; linp.n_joints = 6
    assert int(lib.uvs_rmckf_closed_loop_workspace_bytes(ctypes.byref(mckf), ctypes.byref(linp), 65536)) == 0
    assert int(lib.uvs_rmckf_closed_loop_workspace_bytes(None, ctypes.byref(plant), 65536)) == 0


def test_missing_library_is_loud(uvs, monkeypatch):
    monkeypatch.setattr(uvs._lib, '_lib', None)
    monkeypatch.setattr(uvs._lib, 'LIB_PATH', '/nonexistent/libuvs_rmckf.so')
    with pytest.raises(uvs.UvsLibraryError):
        uvs._lib.lib()


def test_no_gpu_means_no_result(uvs):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(uvs.UvsLibraryError):
        uvs.Experiment([0.0] * 6, [0.0] * 8, None, 0.05, 15, 0.2, uvs.SyntheticRobot(), uvs.Method.GMCKF, initial_guess=True,
                       kernel_bw=10, fpi_threshold=0.1, fpi_epoch_max=10, annealing=False).run()
    with pytest.raises(NotImplementedError):
        uvs.Experiment([0.0] * 6, [0.0] * 8, None, 0.05, 15, 0.2, uvs.SyntheticRobot(), uvs.Method.ANALYTICAL).run()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'uncalibrated-visual-servoing_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, re.M), f
                assert 'oracle/' not in text or f == '_lib.py' or 'test infrastructure' in text, f


# ---------------------------------------------------------------------------------------------- reference-format config
def _ref_config():
    return json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_reference.json')))


def test_config_schema_and_validation(uvs):
    cfg = _ref_config()
    assert uvs.batch.load_config(cfg)['estimator']['method'] == 'MCKF'                 # the reference's shipped default
    bad = json.loads(json.dumps(cfg)); del bad['noise']['hold_time']
    with pytest.raises(KeyError):
        uvs.batch.load_config(bad)
    bad = json.loads(json.dumps(cfg)); bad['estimator']['method'] = 'RMCKF'             # the paper's name is not an enum member
    with pytest.raises(ValueError):
        uvs.batch.load_config(bad)
    bad = json.loads(json.dumps(cfg)); bad['noise']['type'] = 'PINK'
    with pytest.raises(ValueError):
        uvs.batch.load_config(bad)


def test_trial_plan_follows_main_py(uvs):
    cfg = _ref_config()
    plan = uvs.batch.plan_trials(cfg)
    assert len(plan) == 12 * 100 and np.allclose(plan.cells, np.linspace(1, 2, 12))     # ALPHA_STABLE sweeps alpha (main.py:105-106)
    assert list(plan.seed[:3]) == [123456, 123457, 123458] and plan.seed[-1] == 123456 + 1199   # global numbering, never reset
    assert plan.cell[99] == 0 and plan.cell[100] == 1 and plan.value[1199] == 2.0
    # jitter: survey Appendix B known answers for the first trial (main.py:132-134, "2 (r - 1)" formula)
    assert plan.q_start[0, 0] == pytest.approx(-0.26971060839006766, abs=1e-15) and plan.q_start[0, 1] == pytest.approx(-0.229612873789818, abs=1e-15)
    assert np.all(plan.q_start[:, 0] <= 0) and np.all(plan.q_start[:, 0] >= -np.pi / 9 - 1e-12)
    jit = load_golden('noise_uniform_jitter')['values']
    assert np.array_equal(plan.q_start[:300, 0], 2 * (jit[:, 0] - 1) * (np.pi / 18))
    cfg['noise']['type'] = 'GAUSSIAN_MIXTURE'
    cfg['noise']['noise_params'] = dict(std=1.0, mean=50.0, rho=0.0)
    assert np.allclose(uvs.batch.plan_trials(cfg).cells, np.linspace(0, 0.2, 12))
    cfg['experiments']['change_q_start'] = False
    assert np.array_equal(uvs.batch.plan_trials(cfg).q_start[7], cfg['experiments']['q_start'])


def test_loop_clock_matches_reference_logs(uvs):
    t = uvs.engine.loop_clock(0.05, 15)
    assert len(t) == 299 and np.array_equal(t, load_golden('closed_gmckf_a1p5')['t'])
    fp = uvs.engine.make_params(8, 6, 'GMCKF', desired=np.arange(8.0))
    assert fp.steps == 299 and fp.k_max == 300 and fp.reg == 1e-6 and fp.anneal_span == 100 and list(fp.desired)[:8] == list(np.arange(8.0))


@pytest.mark.parametrize('name', golden_names('noise_'))
def test_noise_profiler_and_batch_reproduce_reference_streams(uvs, name):
    g = load_golden(name)
    meta = g['meta']
    nt = uvs.NoiseType[meta['noise_type']]
    prof = uvs.NoiseProfiler(meta['m'], nt, seed=meta['seed'], noise_hold=meta['hold'], noise_hold_cnt=meta['hold_cnt'], noise_params=meta['noise_params'])
    buf = prof.getNoise()
    first = buf.copy()
    assert prof.getNoise() is buf                                                        # aliased buffer, like noise.py:118
    rest = np.stack([prof.getNoise().copy() for _ in range(len(g['values']) - 2)])
    assert np.array_equal(first, g['values'][0]) and np.array_equal(rest, g['values'][2:])
    got = uvs.noise_batch(nt, meta['noise_params'], [meta['seed'], meta['seed'] + 1], meta['m'], len(g['values']), meta['hold'], meta['hold_cnt'])
    # uniform / normal / Cauchy / mixtures are bit-identical; the CMS transform may differ in the last bits (array vs scalar libm)
    assert np.allclose(got[0], g['values'], rtol=1e-14, atol=0)
    assert not np.array_equal(got[0], got[1])


def test_trial_noise_assigns_cells_and_seeds(uvs):
    cfg = _ref_config()
    cfg['experiments']['epoch'] = 3
    plan = uvs.batch.plan_trials(cfg)
    K = 20
    out = np.zeros((len(plan), K, 8))
    uvs.batch.trial_noise(cfg, plan, 0, len(plan), K, out)
    for t in (0, 4, 35):
        p = dict(cfg['noise']['noise_params'], alpha=float(plan.value[t]))
        prof = uvs.NoiseProfiler(8, uvs.NoiseType.ALPHA_STABLE, seed=int(plan.seed[t]), noise_params=p)
        ref = np.stack([prof.getNoise().copy() for _ in range(K)])
        assert np.allclose(out[t], ref, rtol=1e-14, atol=0)


def test_cell_summary_matches_matlab_definitions(uvs):
    rng = np.random.default_rng(0)
    stats, status, cell = rng.uniform(1, 2, (40, 3)), np.zeros(40, int), np.repeat([0, 1], 20)
    status[[3, 25]] = 1
    s = uvs.stats.cell_summary(stats, status, cell)
    ok0 = np.delete(np.arange(20), 3)
    assert s[0]['success'] == 19 and s[0]['itae_mean'] == pytest.approx(stats[ok0, 2].mean())
    assert s[0]['itae_std'] == pytest.approx(stats[ok0, 2].std(ddof=1)) and s[1]['iae_median'] == pytest.approx(np.median(np.delete(stats[20:, 1], 5)))


def test_api_surface_keeps_reference_names(uvs):
    assert [m.name for m in uvs.Method] == ['ANALYTICAL', 'KF', 'MCKF', 'IMCCKF', 'GMCKF'] and uvs.Method.GMCKF.value == 5
    assert [s.name for s in uvs.ExperimentStatus] == ['SUCCESS', 'FAIL'] and str(uvs.ExperimentStatus.SUCCESS) == 'ExperimentStatus.SUCCESS'
    assert [n.name for n in uvs.NoiseType] == ['WHITE_NOISE', 'GAUSSIAN_MIXTURE', 'GAUSSIAN_BIMODAL', 'ALPHA_STABLE', 'UNIFORM']
    assert uvs.NoiseType.numberOfGenerators(uvs.NoiseType.GAUSSIAN_BIMODAL) == 3
    assert uvs.gaussianKernel(3.0, 2.0) == np.exp(-0.5 * 9 / 4)
    ex = uvs.Experiment(q_start=[0] * 6, desired_f=[0] * 8, noise_prof=None, t_s=0.05, t_max=15, ibvs_gain=0.2, robot=object(),
                        method=uvs.Method.GMCKF, method_params=dict(initial_guess=True, kernel_bw=10, fpi_threshold=0.1, fpi_epoch_max=9, annealing=True))
    assert ex.kernel_bw == 10 and ex.annealing is True and ex.fpi_epoch_max == 9       # nested-dict convention (experiment.py:29-30)
    assert len(uvs.batch.CSV_COLUMNS) == 41 and uvs.batch.CSV_COLUMNS[:4] == ['experiment_id', 'status', 'rho', 't']


def test_synthetic_plant_matches_oracle_plant(uvs):
    from oracle import plant_ref
    plant = uvs.SyntheticPlant.ur10()
    assert np.array_equal(plant.points, plant_ref.place_discs())
    q = np.array([0.1, -0.4, 1.9, 0.2, -1.5, 0.3])
    Ts, Tr = plant.fkine_all(q), plant_ref.fkine_all(q)
    assert all(np.array_equal(a, b) for a, b in zip(Ts, Tr)) and np.array_equal(plant.jacobian(Ts), plant_ref.geometric_jacobian(Tr))
    s = plant.to_struct()
    assert s.n_joints == 6 and s.n_points == 4 and s.kind == 0 and s.cos_alpha[0] == np.cos(-np.pi / 2)
    robot = uvs.SyntheticRobot(dt=0.05)
    robot.start(q)
    assert robot.sim.getSimulationTime() == 0.05 and np.array_equal(robot.features(), plant_ref.project(Tr[5], plant.points))


def test_vectorised_pcg64_seeding_matches_numpy(uvs):
    from numpy.random import PCG64
    rng = np.random.default_rng(0)
    seeds = np.concatenate([[0, 1, 12345, 123456, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 3, 2 ** 63 + 11], rng.integers(0, 2 ** 31, 100),
                            rng.integers(2 ** 32, 2 ** 62, 30)]).astype(np.uint64)
    states = uvs.pcg.pcg64_states(seeds)
    for seed, row in zip(seeds, states):
        ref = PCG64(int(seed)).state['state']
        assert (int(row[0]) << 64 | int(row[1])) == ref['state'] and (int(row[2]) << 64 | int(row[3])) == ref['inc']


def test_generator_seed_order_follows_noise_py(uvs):
    NT = uvs.NoiseType
    s = uvs.noise_device.generator_seeds(NT.GAUSSIAN_BIMODAL, [1000, 1001], 8)
    prof = uvs.NoiseProfiler(8, NT.GAUSSIAN_BIMODAL, seed=1000, noise_params=dict(std=1.0, mean=5.0, rho=0.1))
    assert s.shape == (2, 32) and list(s[0, :24]) == [1000 + 10 * j for j in range(24)] and list(s[0, 24:]) == [2000 + i for i in range(8)]
    got = uvs.pcg.pcg64_states(s[0])
    for j, g in enumerate(prof.generators + prof.rhoGenerators):
        st = g.bit_generator.state['state']
        assert (int(got[j, 0]) << 64 | int(got[j, 1])) == st['state'] and (int(got[j, 2]) << 64 | int(got[j, 3])) == st['inc']
    assert uvs.noise_device.generator_seeds(NT.ALPHA_STABLE, [5], 8).shape == (1, 8)
    q = uvs.noise_device.make_noise_params(NT.ALPHA_STABLE, dict(alpha=1.5, beta=0, gamma=1, delta=0), 8, 299, True, 10)
    assert q.hold_cnt == 10 and q.inv_alpha == 1 / 1.5 and q.expo == (1 - 1.5) / 1.5 and q.type == 4
    z = np.load(os.path.join(ROOT, 'uncalibrated-visual-servoing_amd', 'data', 'ziggurat_normal.npz'))
    assert z['fi'][0] == 1.0 and z['ki'][1] == 0 and float(z['wi'][0]) == 8.68362706080130616677e-16 and len(z['ki']) == 256


def test_null_noise_seed_gives_every_trial_its_own_stream():
    """noise.seed: null is legal in the reference (main.py:138 skips the increment, NoiseProfiler(seed=None) draws OS entropy per
    generator): trials must NOT share one sentinel seed, and the seeds must be usable by both generators (seed + 10 j, 2 seed + i < 2^64)."""
    import json
    import os
    import uvs_amd
    from conftest import GOLDEN
    cfg = json.load(open(os.path.join(GOLDEN, 'config_reference.json')))
    cfg['noise']['seed'] = None
    cfg['experiments']['epoch'] = 3
    plan = uvs_amd.batch.plan_trials(cfg, cells=[1.5, 2.0])
    assert len(set(plan.seed.tolist())) == len(plan) == 6 and plan.seed.min() >= 0 and int(plan.seed.max()) < 2 ** 62
    again = uvs_amd.batch.plan_trials(cfg, cells=[1.5, 2.0])
    assert set(again.seed.tolist()).isdisjoint(plan.seed.tolist())          # fresh entropy per plan, like default_rng()
    noise = np.zeros((6, 20, 8))
    uvs_amd.batch.trial_noise(cfg, plan, 0, 6, 20, noise)
    flat = noise.reshape(6, -1)
    assert all(not np.array_equal(flat[a], flat[b]) for a in range(6) for b in range(a + 1, 6))
    gs = uvs_amd.noise_device.generator_seeds(uvs_amd.NoiseType.GAUSSIAN_MIXTURE, plan.seed, 8)
    assert gs.dtype == np.uint64 and len(np.unique(gs)) == gs.size


def test_camera_pose_matches_quat2euler_convention():
    """computePose = position + utils.quat2euler(scalar-first quaternion) (ur10_simulation.py:151-163)."""
    import uvs_amd
    plant = uvs_amd.SyntheticPlant.ur10()
    rng = np.random.default_rng(3)
    for _ in range(20):
        T = plant.fkine_all(rng.uniform(-1.0, 1.0, 6))[-1]
        R = T[:3, :3]
        w = 0.5 * np.sqrt(max(1e-300, 1 + np.trace(R)))
        if w < 0.1:
            continue
        quat = np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
        pose = uvs_amd.plant.camera_pose(T)
        assert np.allclose(pose[:3], T[:3, 3]) and np.allclose(pose[3:], uvs_amd.utils.quat2euler(quat), atol=1e-12)
    robot = uvs_amd.SyntheticRobot(plant)
    robot.start([0.1, -0.2, 1.9, 0.0, -1.5, 0.3])
    assert np.allclose(robot.computePose(), uvs_amd.plant.camera_pose(plant.fkine_all(robot.q)[-1]))


def test_default_kernels_have_no_scratch():
    """The register-bound closed-loop kernels live or die by their allocation (a scratch round trip in the step loop once cost the MCKF
    instantiation 20 %): the library-default (8,6) two-lane instantiations of every estimator, the wide-shape kernel's RMCKF / KF record
    variants and the default replay mappings must not carry a private segment.  Read from the code objects embedded in the built .so."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import kernel_resources
    if not os.path.exists(kernel_resources.READELF):
        pytest.skip('llvm-readelf not available')
    k = kernel_resources.kernels()
    assert len(k) > 100
    for method in (2, 3, 4, 5):                                              # KF, MCKF, IMCCKF, GMCKF
        for plant in (0, 1, 2):                                              # DH, linear, DH with the UR10-like table's compile-time zeros
            for xout in ('true', 'false'):
                r = k[f'closed_loop_tuned_kernel<8, 6, 2, {method}, {plant}, 2, {xout}, false, {"true" if method == 3 else "false"}, false, false, false>']
                # MCKF (round 5): its fixed-point branch -- rare, spread over the wavefront -- runs at the edge of the 512 registers and keeps a few spill
                # slots (a handful of scratch accesses per firing of the branch, none on the plain step: tools/main_path.py); every other
                # estimator carries no private segment at all
                assert r['scratch'] <= (128 if method == 3 else 0) and r['vgpr'] <= 512, (method, plant, xout, r)
                # KF / IMCC-KF keep one covariance block per lane and must fit two wavefronts per SIMD (256 registers, 8 x 19 KB of LDS per CU)
                if method in (2, 4):
                    assert r['vgpr'] <= 256 and r['lds'] <= 20480, (method, r)
    # the small-batch kernels: four lanes per filter with the two-lane kernel's bits (EMU2, the default up to 16 384 trials) and the plain
    # four-lane kernels of UVS_OPT_LATENCY -- one wavefront per SIMD, no scratch (round 3's four-lane RMCKF carried 36-68 B at two per SIMD)
    for method in (2, 4, 5):
        for plant in (0, 2):
            for xout in ('true', 'false'):
                assert k[f'closed_loop_tuned_kernel<8, 6, 4, {method}, {plant}, 2, {xout}, true, false, false, false, false>']['scratch'] == 0
        for xout in ('true', 'false'):
            assert k[f'closed_loop_tuned_kernel<8, 6, 4, {method}, 0, 2, {xout}, false, false, false, false, false>']['scratch'] == 0
    for plant in (0, 2):                                                     # RMCKF's segmented instantiation (launches that are not a whole number of rounds)
        for xout in ('true', 'false'):
            assert k[f'closed_loop_tuned_kernel<8, 6, 2, 5, {plant}, 2, {xout}, false, true, false, false, false>']['scratch'] == 0
    for plant in (0, 2):                                                     # the strict-mode instantiations (round 6: the certificate compiled in)
        for xout in ('true', 'false'):
            assert k[f'closed_loop_tuned_kernel<8, 6, 2, 3, {plant}, 2, {xout}, false, true, false, true, false>']['scratch'] <= 128
            for method in (2, 4):
                r = k[f'closed_loop_tuned_kernel<8, 6, 2, {method}, {plant}, 2, {xout}, false, false, false, true, false>']
                assert r['scratch'] <= 64 and r['vgpr'] <= 256, (method, plant, xout, r)
    for plant in (0, 2):                                                     # the pair-store instantiations (round 6): KF clean, IMCC-KF a few cold bytes
        assert k[f'closed_loop_tuned_kernel<8, 6, 2, 2, {plant}, 2, true, false, false, true, false, true>']['scratch'] == 0
        r = k[f'closed_loop_tuned_kernel<8, 6, 2, 4, {plant}, 2, true, false, false, true, false, true>']
        assert r['scratch'] <= 32 and r['vgpr'] <= 256, r
    for name in ('closed_loop_wide_kernel<32, 7, 8, 5, true, true, 1>', 'closed_loop_wide_kernel<32, 7, 8, 2, true, true, 1>', 'closed_loop_wide_kernel<8, 6, 8, 5, true, false, 0>',
                 'replay_rows_kernel<8, 6, 4, 5, true, true, true, false, 0>', 'replay_rows_kernel<8, 6, 4, 5, true, true, true, false, 2>'):
        assert k[name]['scratch'] == 0, (name, k[name])


def test_noise_kernel_variant_gate(uvs):
    """Which alpha-stable parameters take the beta = 0 instantiation (cos((1 - alpha) V) by the addition theorem): the launcher's error bound
    |(1 - alpha) / alpha| * 2e-16 / cos(|1 - alpha| pi / 2) <= 2e-14, never the reference's special cases (noise.py:180-185)."""
    import ctypes as C
    nd, NT = uvs.noise_device, uvs.NoiseType

    def variant(kind, **p):
        q = nd.make_noise_params(kind, p, 8, 10)
        return uvs.lib().uvs_noise_kernel_variant(C.byref(q))

    S = dict(beta=0.0, gamma=1.0, delta=0.0)
    for alpha in (0.3, 0.5, 0.75, 1.0909090909090908, 1.5, 1.9090909090909092, 1.99):
        assert variant(NT.ALPHA_STABLE, alpha=alpha, **S) == 1, alpha
    for alpha in (0.05, 1.0, 2.0, 1.9999):                                  # error bound / Cauchy / Gaussian special cases
        assert variant(NT.ALPHA_STABLE, alpha=alpha, **S) == 0, alpha
    assert variant(NT.ALPHA_STABLE, alpha=1.5, beta=0.5, gamma=1.0, delta=0.0) == 0
    assert variant(NT.ALPHA_STABLE, alpha=0.5, beta=1.0, gamma=1.0, delta=0.0) == 0
    assert variant(NT.WHITE_NOISE, std=1.0) == 0 and variant(NT.GAUSSIAN_MIXTURE, std=1.0, mean=50.0, rho=0.1) == 0
    # the bound itself, on both sides of its two edges
    lo = [a for a in np.linspace(0.05, 0.3, 251) if variant(NT.ALPHA_STABLE, alpha=float(a), **S) == 1][0]
    hi = [a for a in np.linspace(1.99, 2.0, 1001)[:-1] if variant(NT.ALPHA_STABLE, alpha=float(a), **S) == 1][-1]
    for a in (lo, hi):
        assert abs((1 - a) / a) * 2e-16 <= 2e-14 * np.cos(abs(1 - a) * np.pi / 2)
    assert 0.05 < lo < 0.2 and 1.995 < hi < 2.0



@pytest.mark.parametrize('base,shifted', [('noise_alpha1p5', 'noise_alpha1p5_seed_plus10'), ('noise_alpha1p0', 'noise_alpha1p0_seed_plus10'),
                                          ('noise_white', 'noise_white_seed_plus10'), ('noise_uniform_jitter', 'noise_uniform_jitter_seed_plus10')])
def test_reference_streams_alias_across_trials(uvs, base, shifted):
    """What the sweep's shared noise buffer rests on, pinned on the reference's own output (oracle/gen_golden.py ran NoiseProfiler at seed and
    at seed + 10): generator i is PCG64(seed + 10 i) (noise.py:70), so feature i + 1 of the trial seeded s is, bit for bit, feature i of the
    trial seeded s + 10 -- and the driver seeds trial t with seed + t (main.py:137-139)."""
    a, b = load_golden(base), load_golden(shifted)
    assert b['meta']['seed'] == a['meta']['seed'] + 10 and not a['meta']['hold'] and a['meta']['noise_params'] == b['meta']['noise_params']
    assert np.array_equal(a['values'][:, 1:], b['values'][:, :-1]) and not np.array_equal(a['values'][:, 0], b['values'][:, 0])
    nt = uvs.NoiseType[a['meta']['noise_type']]
    seeds = a['meta']['seed'] + np.arange(11)
    assert uvs.noise_device.shares_streams(nt, False, seeds)
    assert not uvs.noise_device.shares_streams(nt, True, seeds)                        # the hold couples the two features of a pair (noise.py:82-116)
    assert not uvs.noise_device.shares_streams(nt, False, seeds[::2])                  # only consecutive seeds line up with the stride 10
    # the host generator, walked the same way: [step][S] streams, entry (k, i, t) of the dense batch = stream t + 10 i
    m, K = a['meta']['m'], 40
    dense = uvs.noise_batch(nt, a['meta']['noise_params'], seeds, m, K)                   # (T, K, m)
    S = len(seeds) + 10 * (m - 1)
    streams = uvs.noise_batch(nt, a['meta']['noise_params'], a['meta']['seed'] + np.arange(S), m, K)[:, :, 0]     # feature 0 of S trials
    for i in range(m):
        assert np.array_equal(dense[:, :, i], streams[10 * i:10 * i + len(seeds)])


def test_sweep_pieces_cover_a_shard_cell_by_cell():
    """batch.sweep_pieces: a shard's trials in order, cut at the cell boundaries (main.py:121-127) and at max_trials."""
    from uvs_amd import batch, dist
    cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_reference.json')))
    cfg['experiments']['epoch'] = 150
    plan = batch.plan_trials(cfg)                                  # 12 cells x 150 trials
    for world in (1, 4, 7):
        for rank in range(world):
            lo, hi = dist.shard_range(len(plan), rank, world)
            for cap in (None, 64, 150, 1000):
                pieces = batch.sweep_pieces(plan, lo, hi, cap)
                assert [a for a, _, _ in pieces] == [lo] + [b for _, b, _ in pieces[:-1]] and pieces[-1][1] == hi
                for a, b, c in pieces:
                    assert a < b and set(plan.cell[a:b]) == {c} and (cap is None or b - a <= cap)
                if cap is None or cap >= 150:                        # whole cells: one piece per cell the shard touches
                    assert len(pieces) == len(set(plan.cell[lo:hi]))
    assert batch.sweep_pieces(plan, 5, 5) == []


def test_counter_figures_belong_to_this_library(uvs):
    """profiles/traffic_latest.json feeds bench.py's roofline.traffic / roofline.valu with PMC counts of an EARLIER profiling call.  They are valid
    only for the kernels they were counted on: the file records the fingerprint of the kernel sources it was taken with (csrc/src_hash.py:
    comments and blank space stripped), the library carries the same fingerprint in uvs_version(), and this test fails when the sources, the
    built library and the counter file do not all agree -- so the counts cannot outlive a kernel change (VERDICT r5 #6b)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('src_hash', os.path.join(ROOT, 'uncalibrated-visual-servoing_amd', 'csrc', 'src_hash.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    version = uvs.lib().uvs_version().decode()
    assert version.endswith('src:' + mod.source_hash()), 'libuvs_rmckf.so is older than its sources: rebuild (make -C uncalibrated-visual-servoing_amd/csrc)'
    tr = json.load(open(os.path.join(ROOT, 'profiles', 'traffic_latest.json')))
    assert tr.get('library_version') == version, ('profiles/traffic_latest.json was counted on other kernels: re-run tools/profile_round.sh + '
                                                  'tools/make_traffic_json.py', tr.get('library_version'), version)
    assert tr['kernel'].startswith('closed_loop_tuned_kernel<8,6,2,GMCKF') and tr['round'] >= 6


def test_documents_stay_readable():
    """No line of the Markdown documents over 160 characters (evidence legibility) -- `python tools/reflow_md.py <file>` re-wraps paragraphs and
    turns a table with an over-long row into a list."""
    for name in ('DESIGN.md', 'DESIGN_APPENDIX.md', 'README.md', 'INTEGRATION.md', os.path.join('profiles', 'README.md')):
        longest = max(len(line) for line in open(os.path.join(ROOT, name), encoding='utf-8').read().split('\n'))
        assert longest <= 160, (name, longest)

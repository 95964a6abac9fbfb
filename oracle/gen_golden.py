"""Generate tests/golden/*.npz by running the UNMODIFIED reference in the build container.

BUILD-CONTAINER ONLY: imports /root/reference (read-only) and therefore never
runs on the GPU box; only the .npz vectors it writes travel.  Recipe follows
SURVEY.md Appendix A: stub ``cv2`` (imported by utils.py:2, used only inside the
detectors) and the ZMQ client (ur10_simulation.py:3), subclass the reference's
``UR10Simulation`` so that its own ``fkine``/``jacobian``/``dh`` are the plant
kinematics, bind ``experiment.detect4Circles`` to the plant's pinhole
projection, and capture the estimator state with ``sys.settrace`` at
experiment.py:302 (first line after the filter update).

    python oracle/gen_golden.py            # rewrites tests/golden/
    python oracle/gen_golden.py noise_alpha1p5_seed_plus10 ...   # only the named fixtures

Every fixture stores inputs (q_start, noise stream actually drawn, parameters)
and the reference's outputs (status, t/err/q/f logs, per-step X, selected
covariance blocks).
"""
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

sys.path.insert(0, REF)
sys.modules.setdefault('cv2', types.ModuleType('cv2'))
_zmq = types.ModuleType('coppeliasim_zmqremoteapi_client')
_zmq.RemoteAPIClient = object
sys.modules.setdefault('coppeliasim_zmqremoteapi_client', _zmq)

import experiment as E                                                # noqa: E402  (reference)
import ur10_simulation as U                                           # noqa: E402  (reference)
from noise import NoiseProfiler, NoiseType                           # noqa: E402  (reference)

ONLY = set(sys.argv[1:])                                              # fixture names to (re)write; none = all
DT, T_MAX, GAIN = 0.05, 15, 0.2
FOCAL = 256 / (2 * np.tan(0.5 * np.deg2rad(65)))
DESIRED = np.array([149.0, 145.0, 125.0, 121.0, 101.0, 145.0, 125.0, 169.0])
Q_START = np.array([0.0, 0.0, 1.96349541, 0.0, -1.57079633, 0.0])
Q_GOAL = np.array([0.0, -np.pi / 8, np.pi / 2 + np.pi / 8, 0.0, -np.pi / 2, 0.0])
P_STEPS = (0, 1, 2, 5, 10, 50, 100, 200, 298)


class _Clock:
    t = 0.0

    def getSimulationTime(self):
        return self.t


class RefPlant(U.UR10Simulation):
    """Reference kinematics + pinhole camera + kinematic joints (Appendix A)."""

    def __init__(self):
        self.q = Q_GOAL.copy()
        self.qt = self.q.copy()
        self.perspective_angle = 65
        self.sim = _Clock()
        T = self.fkine(recalculate=True)
        d = T[2, 3]
        self.discs = [T[:3, 3] + T[:3, :3] @ np.array([(DESIRED[2 * i] - 128) / FOCAL * d, (DESIRED[2 * i + 1] - 128) / FOCAL * d, d])
                      for i in range(4)]

    def __del__(self):
        pass

    def start(self, q):
        self.q = np.array(q, float)
        self.qt = self.q.copy()
        self.fkine(recalculate=True, all_transforms=True)
        self.step()

    def stop(self):
        pass

    def step(self):
        self.q = self.qt.copy()
        self.sim.t += DT

    def getJointsPos(self):
        return self.q

    def setJointsPos(self, q):
        self.qt = np.array(q, float)

    def computePose(self, recalculate_fkine=False):
        return np.r_[self.fkine(recalculate=True)[:3, 3], 0, 0, 0]

    def computeZ(self, n=1, recalculate_fkine=False):
        c = self.getCameraPosition(recalculate_fkine)
        return np.array([np.linalg.norm(c - d) for d in self.discs[:n]])

    def getCameraImage(self):
        return self, (256, 256)

    def features(self):
        T = self.fkine(recalculate=True)
        R, t = T[:3, :3], T[:3, 3]
        f = np.zeros(8)
        for i, d in enumerate(self.discs):
            pc = R.T @ (d - t)
            f[2 * i:2 * i + 2] = 128 + FOCAL * pc[0] / pc[2], 128 + FOCAL * pc[1] / pc[2]
        return f


E.detect4Circles = lambda image: image.features()                     # name bound at experiment.py:2


def run_reference(method, noise_type, noise_params, seed, q_start=Q_START, hold=False, hold_cnt=10, **mp):
    rec, code = [], E.Experiment.run.__code__

    def local(frame, event, arg):
        if event == 'line' and frame.f_lineno == 302:
            L = frame.f_locals
            rec.append({k: np.array(L[k], copy=True) for k in ('X', 'P', 'dq', 'e', 'kernel_bw', 'epoch', 'skip_correction') if k in L})
        return local

    npf = NoiseProfiler(num_features=8, noise_type=noise_type, seed=seed, noise_hold=hold, noise_hold_cnt=hold_cnt,
                        noise_params=dict(noise_params))
    params = dict(initial_guess=True, kernel_bw=10, fpi_threshold=0.1, fpi_epoch_max=1000, annealing=False)
    params.update(mp)
    ex = E.Experiment(q_start=np.array(q_start, float), desired_f=DESIRED, noise_prof=npf, t_s=DT, t_max=T_MAX, ibvs_gain=GAIN,
                      robot=RefPlant(), method=method, method_params=params)
    sys.settrace(lambda fr, ev, arg: local if fr.f_code is code else None)
    try:
        out = ex.run()
    finally:
        sys.settrace(None)
    return out, rec, params


def save_closed(name, method, noise_type, noise_params, seed, x_stride=1, prefix='closed_', f_init=None, extra=None, **kw):
    if ONLY and f'{prefix}{name}' not in ONLY:
        return
    out, rec, params = run_reference(method, noise_type, noise_params, seed, **kw)
    status, t, err, q, f, fd, cam, noise, bw = out
    k = len(t)
    X = np.stack([r['X'].ravel() for r in rec])[:k]
    steps = [s for s in P_STEPS if s < k]
    P_blocks = np.stack([np.stack([rec[s]['P'][6 * i:6 * i + 6, 6 * i:6 * i + 6] for i in range(8)]) for s in steps])
    mask = np.kron(np.eye(8), np.ones((6, 6))) == 0
    off_block = max(float(np.abs(r['P'][mask]).max()) for r in rec)
    x_idx = np.unique(np.r_[np.arange(0, k, x_stride), np.arange(min(k, 12)), k - 1])
    if f_init is None:
        plant0 = RefPlant()
        plant0.start(np.array(kw.get('q_start', Q_START), float))
        f_init = plant0.features()                                    # f before the loop (experiment.py:90)
    dq_prev = np.stack([r['dq'].ravel() for r in rec])[:k]            # regressor used at step k (experiment.py:188)
    meta = dict(method=method.name, noise_type=noise_type.name, noise_params=noise_params, seed=seed,
                hold=bool(kw.get('hold', False)), hold_cnt=int(kw.get('hold_cnt', 10)), params=params,
                dt=DT, t_max=T_MAX, gain=GAIN, generator='oracle/gen_golden.py', reference='experiment.py Experiment.run()')
    np.savez_compressed(os.path.join(OUT, f'{prefix}{name}.npz'), **(extra or {}),
                        meta=json.dumps(meta), status=status.value, q_start=np.array(kw.get('q_start', Q_START), float), desired=DESIRED,
                        t=t, err=err, q=q, f=f, noise=noise, cam=cam, sigma_log=bw, f_init=f_init, dq_prev=dq_prev,
                        X=X[x_idx], X_steps=x_idx, P_steps=np.array(steps), P_blocks=P_blocks, P_offblock_max=off_block,
                        e=np.stack([r['e'].ravel() for r in rec])[:k] if 'e' in rec[0] else np.zeros(0),
                        sigma=np.array([float(r['kernel_bw']) for r in rec])[:k] if 'kernel_bw' in rec[0] else np.zeros(0))
    print(f'{prefix}{name}: status={status.name} k={k} |err[-1]|={np.linalg.norm(err[-1]):.6g} offblock={off_block:g}')
    return out, rec


def save_noise(name, m, noise_type, noise_params, seed, calls=300, hold=False, hold_cnt=10):
    if ONLY and f'noise_{name}' not in ONLY:
        return
    npf = NoiseProfiler(num_features=m, noise_type=noise_type, seed=seed, noise_hold=hold, noise_hold_cnt=hold_cnt,
                        noise_params=dict(noise_params))
    vals = np.stack([npf.getNoise().copy() for _ in range(calls)])
    meta = dict(noise_type=noise_type.name, noise_params=noise_params, seed=seed, hold=hold, hold_cnt=hold_cnt, m=m,
                generator='oracle/gen_golden.py', reference='noise.py NoiseProfiler.getNoise()')
    np.savez_compressed(os.path.join(OUT, f'noise_{name}.npz'), meta=json.dumps(meta), values=vals)
    print(f'noise_{name}: first={vals[0, :3]} absmax={np.abs(vals).max():.4g}')


def main():
    os.makedirs(OUT, exist_ok=True)
    AS = lambda a, b=0, g=1, d=0: dict(alpha=a, beta=b, gamma=g, delta=d)      # noqa: E731
    MIX = dict(std=1.0, mean=50.0, rho=0.1)
    M, NT = E.Method, NoiseType

    # -- noise streams (row N of SURVEY 8a) ---------------------------------
    save_noise('white', 8, NT.WHITE_NOISE, dict(std=1.0), 123456)
    save_noise('mixture', 8, NT.GAUSSIAN_MIXTURE, MIX, 123456)
    save_noise('mixture_hold', 8, NT.GAUSSIAN_MIXTURE, MIX, 123456, hold=True)
    save_noise('bimodal', 8, NT.GAUSSIAN_BIMODAL, MIX, 123457)
    save_noise('bimodal_hold', 8, NT.GAUSSIAN_BIMODAL, dict(std=2.0, mean=30.0, rho=0.3), 99, hold=True, hold_cnt=4)
    save_noise('alpha2p0', 8, NT.ALPHA_STABLE, AS(2.0), 123456)
    save_noise('alpha1p0', 8, NT.ALPHA_STABLE, AS(1.0), 123456)
    save_noise('alpha1p5', 8, NT.ALPHA_STABLE, AS(1.5), 123456)
    save_noise('alpha1p5_hold', 8, NT.ALPHA_STABLE, AS(1.5), 123460, hold=True)
    save_noise('alpha1p0_hold', 8, NT.ALPHA_STABLE, AS(1.0), 123461, hold=True)
    save_noise('alpha0p5_levy', 8, NT.ALPHA_STABLE, AS(0.5, 1), 7)
    save_noise('alpha1p2_beta0p5', 8, NT.ALPHA_STABLE, AS(1.2, 0.5, 2.0, 1.0), 123462)
    save_noise('alpha1p0_beta0p5', 8, NT.ALPHA_STABLE, AS(1.0, 0.5, 2.0, 1.0), 123463)
    save_noise('alpha1p0909', 8, NT.ALPHA_STABLE, AS(float(np.linspace(1, 2, 12)[1])), 123556)
    save_noise('uniform_jitter', 2, NT.UNIFORM, {}, 12345)
    save_noise('white_m2', 2, NT.WHITE_NOISE, dict(std=1.0), 123456)
    # the Monte-Carlo driver's next-but-nine trial (main.py:137-139: seed + trial): generator i of a profiler is PCG64(seed + 10 i)
    # (noise.py:70), so feature i of these IS feature i + 1 of the fixtures at seed 123456 / 12345 -- the aliasing the sweep exploits
    save_noise('alpha1p5_seed_plus10', 8, NT.ALPHA_STABLE, AS(1.5), 123466)
    save_noise('alpha1p0_seed_plus10', 8, NT.ALPHA_STABLE, AS(1.0), 123466)
    save_noise('white_seed_plus10', 8, NT.WHITE_NOISE, dict(std=1.0), 123466)
    save_noise('uniform_jitter_seed_plus10', 2, NT.UNIFORM, {}, 12355)

    # -- closed loop, GMCKF (= the paper's RMCKF): rows S0-S11 ----------------
    save_closed('gmckf_a1p5', M.GMCKF, NT.ALPHA_STABLE, AS(1.5), 123456)
    save_closed('gmckf_a1p5_anneal', M.GMCKF, NT.ALPHA_STABLE, AS(1.5), 123456, annealing=True)
    save_closed('gmckf_a1p0', M.GMCKF, NT.ALPHA_STABLE, AS(1.0), 123457, x_stride=8)
    save_closed('gmckf_a2p0', M.GMCKF, NT.ALPHA_STABLE, AS(2.0), 123458, x_stride=8)
    save_closed('gmckf_white', M.GMCKF, NT.WHITE_NOISE, dict(std=1.0), 123456, x_stride=8)
    save_closed('gmckf_mix_anneal', M.GMCKF, NT.GAUSSIAN_MIXTURE, MIX, 123456, annealing=True)
    save_closed('gmckf_mix_anneal_hold', M.GMCKF, NT.GAUSSIAN_MIXTURE, MIX, 123459, annealing=True, hold=True, x_stride=8)
    save_closed('gmckf_a1p5_hold', M.GMCKF, NT.ALPHA_STABLE, AS(1.5), 123460, hold=True, x_stride=8)
    save_closed('gmckf_bimodal', M.GMCKF, NT.GAUSSIAN_BIMODAL, MIX, 123461, x_stride=8)
    save_closed('gmckf_sigma1', M.GMCKF, NT.ALPHA_STABLE, AS(2.0), 123456, kernel_bw=1, x_stride=8)
    # jittered start: first pair of the UNIFORM stream seeded 12345, main.py:132-134
    jit = NoiseProfiler(num_features=2, noise_type=NT.UNIFORM, seed=12345).getNoise().copy()
    qj = Q_START.copy()
    qj[0] += 2 * (jit[0] - 1) * (np.pi / 18)
    qj[1] += 2 * (jit[1] - 1) * (np.pi / 9)
    save_closed('gmckf_a1p5_jitter', M.GMCKF, NT.ALPHA_STABLE, AS(1.5), 123456, q_start=qj, x_stride=8)

    # -- the other estimators ------------------------------
    for meth in (M.KF, M.IMCCKF, M.MCKF):
        save_closed(f'{meth.name.lower()}_a2p0', meth, NT.ALPHA_STABLE, AS(2.0), 123456, x_stride=8)
        save_closed(f'{meth.name.lower()}_a1p5', meth, NT.ALPHA_STABLE, AS(1.5), 123456, x_stride=8)


if __name__ == '__main__':
    main()

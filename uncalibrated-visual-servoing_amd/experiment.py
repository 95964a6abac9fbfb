"""``Experiment`` with the reference's interface (experiment.py:6-359), estimator on the MI355X.

Same enums, constructor signature (flat kwargs or a nested ``method_params=`` dict, experiment.py:29-30) and
``run()`` 9-tuple ``(status, t_log, error_log, q_log, f_log, desired_f_log, camera_log, noise_log, kernel_bw_log)``.

Two execution routes, both through libuvs_rmckf (never numpy):
  * robot is a ``SyntheticRobot``: the whole trial -- plant, noise, estimator, control law, logs -- is one launch of the
    closed-loop kernel with T = 1 (the Monte-Carlo driver in ``batch.py`` launches the same kernel with T in the 10^4..10^6);
  * any other duck-typed robot (e.g. the reference's ``UR10Simulation`` on a live CoppeliaSim): the Python loop keeps the
    reference's call order on the robot, and each estimator + control-law step runs on the GPU with the filter state
    resident in HBM (``FilterBank``).
"""
import logging
from enum import Enum

import numpy as np

from . import engine
from .plant import SyntheticRobot, camera_pose
from .utils import gaussianKernel  # noqa: F401  (re-exported like the reference module namespace)


class Method(Enum):
    ANALYTICAL = 1
    KF = 2
    MCKF = 3
    IMCCKF = 4
    GMCKF = 5


class ExperimentStatus(Enum):
    SUCCESS = 0
    FAIL = 1


def detect4Circles(image):
    """Feature extraction hook bound in this module's namespace like the reference (experiment.py:2, utils.py:126-166).
    Synthetic robots hand over an object that already knows its features; a camera frame (H x W x 3 uint8, as
    UR10Simulation.getCameraImage returns it) goes through the centre-of-mass detector."""
    if hasattr(image, 'features'):
        return image.features()
    from .utils import detect4Circles as _detect
    return _detect(image)


_GPU_METHODS = (Method.KF, Method.MCKF, Method.IMCCKF, Method.GMCKF)


def bandwidth_log(method, k, kernel_bw, annealing, t_s, t_max):
    """kernel_bw_log of a k-row trial: the (annealed) bandwidth for MCKF, -1 for every other method -- GMCKF included
    (experiment.py:330; the MCKF branch rebinds kernel_bw per step at :196-200, every other branch leaves the initial -1)."""
    if method != Method.MCKF:
        return np.full(k, -1.0)
    k_max = int(t_max / t_s)                                         # experiment.py:120
    steps = np.arange(k)
    return (kernel_bw + 100 * (1 - steps / k_max)) if annealing else np.full(k, float(kernel_bw))


class Experiment:
    def __init__(self, q_start: list, desired_f: list, noise_prof: object, t_s: float, t_max: float, ibvs_gain: float,
                 robot: object, method: Method, logger: object = None, **method_params) -> None:
        self.q_start, self.desired_f, self.robot, self.noise_prof = q_start, desired_f, robot, noise_prof
        self.t_s, self.t_max, self.ibvs_gain, self.method = t_s, t_max, ibvs_gain, method
        if 'method_params' in method_params:
            method_params = method_params['method_params']
        if method in (Method.KF, Method.MCKF, Method.IMCCKF, Method.GMCKF):
            self.initial_guess = method_params['initial_guess']
            if method != Method.KF:
                self.kernel_bw = method_params['kernel_bw']
                self.fpi_threshold = method_params['fpi_threshold']
                self.fpi_epoch_max = method_params['fpi_epoch_max']
                self.annealing = method_params['annealing']
        self.lanes = int(method_params.get('lanes_per_filter', 0))
        self.x0 = method_params.get('x0')            # explicit initial state when initial_guess is False (reference: unseeded random)
        self.logger = logging.getLogger(__name__)
        if logger is not None:
            self.logger.setLevel(logger.level)

    # ------------------------------------------------------------------------------------------------
    def _params(self, m, n, steps=None):
        return engine.make_params(m, n, self.method.name, getattr(self, 'kernel_bw', 1.0), getattr(self, 'annealing', False),
                                  self.t_s, self.t_max, self.ibvs_gain, self.desired_f, self.initial_guess, self.lanes, steps,
                                  getattr(self, 'fpi_threshold', 0.1), getattr(self, 'fpi_epoch_max', 1000))

    def _bandwidth_log(self, k):
        return bandwidth_log(self.method, k, getattr(self, 'kernel_bw', 1.0), getattr(self, 'annealing', False), self.t_s, self.t_max)

    def run(self) -> list:
        if self.method not in _GPU_METHODS:
            raise NotImplementedError(f'{self.method.name} is not an estimator of the HIP path (KF, MCKF, IMCCKF, GMCKF are)')
        if isinstance(self.robot, SyntheticRobot):
            return self._run_on_device_plant()
        return self._run_with_external_robot()

    # ---- route 1: plant inside the kernel ------------------------------------------------------------
    def _run_on_device_plant(self):
        torch = engine._torch()                                  # raises UvsLibraryError without a GPU: there is no CPU fallback
        robot, m = self.robot, len(self.desired_f)
        n = robot.plant.n_joints
        t_log = engine.loop_clock(self.t_s, self.t_max)
        K = len(t_log)
        fp = self._params(m, n)
        noise_log = np.zeros((K, m))
        if self.noise_prof is not None:
            for k in range(K):
                noise_log[k] = self.noise_prof.getNoise()            # aliased buffer in the reference: copy per call
        dev = torch.device('cuda')
        noise = torch.as_tensor(noise_log.reshape(K, m, 1).copy(), device=dev)
        q0 = torch.as_tensor(np.asarray(self.q_start, float).reshape(1, n), device=dev)
        x0 = None
        if not self.initial_guess:
            if self.x0 is None:
                raise ValueError('initial_guess=False needs method_params["x0"] (the reference draws it unseeded, experiment.py:117)')
            x0 = torch.as_tensor(np.asarray(self.x0, float).reshape(1, m * n), device=dev)
        out = engine.closed_loop(fp, robot.plant.to_struct(), q0, noise, x0, want=('err', 'q', 'f'))
        k = int(out['k_done'].item())
        status = ExperimentStatus(int(out['status'].item()))
        err = out['err'][:k, :, 0].cpu().numpy()
        q_log = out['q'][:k, :, 0].cpu().numpy()
        f_log = out['f'][:k, :, 0].cpu().numpy()
        cam = np.zeros((k, 6))
        for i in range(k):
            cam[i] = camera_pose(robot.plant.fkine_all(q_log[i])[-1])          # computePose (ur10_simulation.py:151-163)
        if status == ExperimentStatus.FAIL:
            self.logger.error('Experiment failed')
        else:
            self.logger.info('Experiment success')
        bw_log = self._bandwidth_log(k)
        return (status, t_log[:k], err, q_log, f_log, np.tile(np.asarray(self.desired_f, float), (k, 1)), cam, noise_log[:k], bw_log)

    # ---- route 2: external robot, estimator step on the GPU ------------------------------------------
    def _run_with_external_robot(self):
        torch = engine._torch()
        robot = self.robot
        robot.start(self.q_start)
        m, n = len(self.desired_f), 6
        rows = int(self.t_max / self.t_s)
        logs = dict(err=np.zeros((rows, m)), f=np.zeros((rows, m)), q=np.zeros((rows, 6)), cam=np.zeros((rows, n)), t=np.zeros(rows),
                    des=np.zeros((rows, m)), noise=np.zeros((rows, m)), bw=np.zeros(rows))
        f = np.zeros(m)
        noise = np.zeros(m)
        x0 = None
        if self.initial_guess:
            image, resolution = robot.getCameraImage()
            try:
                f = np.array(detect4Circles(image), float)
            except Exception as exc:                                 # reference: log and continue (experiment.py:91-92)
                self.logger.error(exc)
            x0 = self._analytic_guess(robot, f, resolution, m, n)
        elif self.x0 is not None:
            x0 = np.asarray(self.x0, float)
        else:
            x0 = np.random.default_rng().random(m * n)               # experiment.py:117
        dev = torch.device('cuda')
        bank = engine.FilterBank(self._params(m, n, steps=0), 1, x0, dev)
        dq = np.zeros(n)
        status, k = ExperimentStatus.SUCCESS, 0
        while (t := robot.sim.getSimulationTime()) < self.t_max:
            image, resolution = robot.getCameraImage()
            f_old = f.copy()
            try:
                f = np.array(detect4Circles(image), float)
                if self.noise_prof is not None:
                    noise = self.noise_prof.getNoise()
                    f += noise
            except Exception as exc:
                self.logger.error(exc)
            robot.computePose(recalculate_fkine=True)
            dq_h, err_h, _, st = bank.step_host(f, f_old, k, dq)    # zero-copy: inputs and outputs in pinned host memory, one launch
            if st[0] != 0:                                           # non-finite X: pinv would raise (experiment.py:313-316)
                status = ExperimentStatus.FAIL
                self.logger.error('Experiment failed')
                break
            dq = dq_h[0].copy()
            q_now = robot.getJointsPos()
            new_q = q_now + dq * self.t_s
            logs['q'][k], logs['cam'][k], logs['f'][k], logs['des'][k] = q_now, robot.computePose(), f, self.desired_f
            logs['err'][k], logs['noise'][k], logs['t'][k] = err_h[0], noise, t
            k += 1
            robot.setJointsPos(new_q)
            robot.step()
        robot.stop()
        if status == ExperimentStatus.SUCCESS:
            self.logger.info('Experiment success')
        return (status, logs['t'][:k], logs['err'][:k], logs['q'][:k], logs['f'][:k], logs['des'][:k], logs['cam'][:k],
                logs['noise'][:k], self._bandwidth_log(k))

    @staticmethod
    def _analytic_guess(robot, f, resolution, m, n):
        """X0 for an external robot (experiment.py:94-114), evaluated once on the host from the robot's own kinematics."""
        depth = robot.computeZ(m // 2)
        focal = resolution[0] / (2 * np.tan(0.5 * np.deg2rad(robot.perspective_angle)))
        Ji = np.zeros((m, 6))
        for i in range(m // 2):
            u, v = f[2 * i], f[2 * i + 1]
            Ji[2 * i] = [-focal / depth[i], 0.0, u / depth[i], u * v / focal, -(focal ** 2 + u ** 2) / focal, v]
            Ji[2 * i + 1] = [0.0, -focal / depth[i], v / depth[i], (focal ** 2 + v ** 2) / focal, -u * v / focal, -u]
        return (Ji @ np.kron(np.eye(2), robot.getCameraRotation().T) @ robot.jacobian()).reshape(m * n)

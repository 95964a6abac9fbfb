"""Synthetic plants for closed-loop Monte-Carlo runs: a DH serial arm carrying a pinhole camera.

The reference talks to an external CoppeliaSim process (ur10_simulation.py RPC methods); what it computes
locally is the UR10 DH forward kinematics and geometric Jacobian (ur10_simulation.py:97-139, 204-211).
``SyntheticPlant`` keeps exactly those kinematics and closes the loop with a pinhole camera on the last frame
looking at fixed world points.  The same description is (a) handed to the HIP kernels
as a ``uvs_plant`` struct, where the plant runs inside the closed-loop kernel, and (b) evaluated on the host
by ``SyntheticRobot``, the duck-typed robot that ``Experiment`` drives.
"""
from dataclasses import dataclass, field

import numpy as np

from . import _lib

RESOLUTION = 256
FOV_DEG = 65.0


def focal_length(resolution=RESOLUTION, fov_deg=FOV_DEG):
    return resolution / (2 * np.tan(0.5 * np.deg2rad(fov_deg)))            # experiment.py:97


def camera_pose(T):
    """[x, y, z, roll, pitch, yaw] of a homogeneous camera transform, as UR10Simulation.computePose returns it
    (ur10_simulation.py:151-163): position, then the angles utils.quat2euler extracts from the simulator's scalar-first
    quaternion -- R = Rz(yaw) Ry(pitch) Rx(roll), i.e. roll = atan2(R21, R22), pitch = asin(-R20), yaw = atan2(R10, R00)
    (the reference's own commented-out matrix form, ur10_simulation.py:158-160)."""
    T = np.asarray(T, float)
    pose = np.empty(6)
    pose[:3] = T[:3, 3]
    pose[3] = np.arctan2(T[2, 1], T[2, 2])
    pose[4] = np.arcsin(np.clip(-T[2, 0], -1.0, 1.0))
    pose[5] = np.arctan2(T[1, 0], T[0, 0])
    return pose


@dataclass
class SyntheticPlant:
    theta_offset: np.ndarray
    d: np.ndarray
    a: np.ndarray
    alpha: np.ndarray
    points: np.ndarray                      # (n_points, 3) world coordinates
    focal: float = field(default_factory=focal_length)
    center: float = RESOLUTION / 2
    fov_deg: float = FOV_DEG

    # ---------------------------------------------------------------- constructors
    @classmethod
    def ur10(cls, desired_f=(149.0, 145.0, 125.0, 121.0, 101.0, 145.0, 125.0, 169.0), q_goal=None):
        """UR10 chain of ur10_simulation.py:100-105; discs on the floor that project onto ``desired_f`` (config.json:9)
        from the straight-down pose used by the reference's tests/*.py."""
        half = np.pi / 2
        plant = cls(theta_offset=np.array([0.0, -half, 0.0, -half, 0.0, np.pi]),
                    d=np.array([0.128, 0.0, 0.0, 0.1639, 0.1157, 0.0922]),
                    a=np.array([0.0, 0.6127, 0.5716, 0.0, 0.0, 0.0]),
                    alpha=np.array([-half, 0.0, 0.0, -half, half, 0.0]),
                    points=np.zeros((len(desired_f) // 2, 3)))
        q_goal = np.array([0.0, -np.pi / 8, half + np.pi / 8, 0.0, -half, 0.0]) if q_goal is None else np.asarray(q_goal, float)
        T = plant.fkine_all(q_goal)[-1]
        depth = T[2, 3]
        for i in range(len(desired_f) // 2):
            ray = np.array([(desired_f[2 * i] - plant.center) / plant.focal * depth,
                            (desired_f[2 * i + 1] - plant.center) / plant.focal * depth, depth])
            plant.points[i] = T[:3, 3] + T[:3, :3] @ ray
        return plant

    # ---------------------------------------------------------------- host kinematics
    @property
    def n_joints(self):
        return len(self.d)

    @property
    def n_points(self):
        return len(self.points)

    def link(self, i, theta):
        c, s = np.cos(theta), np.sin(theta)
        ca, sa = np.cos(self.alpha[i]), np.sin(self.alpha[i])
        rz = np.array([[c, -s, 0.0, 0.0], [s, c, 0.0, 0.0], [0.0, 0.0, 1.0, self.d[i]], [0.0, 0.0, 0.0, 1.0]])
        rx = np.array([[1.0, 0.0, 0.0, self.a[i]], [0.0, ca, -sa, 0.0], [0.0, sa, ca, 0.0], [0.0, 0.0, 0.0, 1.0]])
        return rz @ rx                                              # ur10_simulation.py:204-211

    def fkine_all(self, q):
        Ts, T = [], None
        for i in range(self.n_joints):
            A = self.link(i, q[i] + self.theta_offset[i])
            T = A if T is None else T @ A
            Ts.append(T)
        return Ts

    def jacobian(self, Ts):
        p_e = Ts[-1][:3, 3]
        z, p, cols = np.array([0.0, 0.0, 1.0]), np.zeros(3), []
        for T in Ts:
            cols.append(np.concatenate([np.cross(z, p_e - p), z]))  # ur10_simulation.py:132-137
            z, p = T[:3, 2], T[:3, 3]
        return np.stack(cols, axis=1)

    def camera_pose_batch(self, q):
        """camera_pose(fkine(q)) for a whole array of joint vectors, q of shape (..., n) -> (..., 6): the same link products as
        ``link`` / ``fkine_all`` (ur10_simulation.py:97-110, 204-211) evaluated with stacked 4 x 4 matrices, for the sinks that log the
        camera pose of every step of every trial (results.csv / Parquet, main.py:160-165)."""
        q = np.asarray(q, float)
        T = None
        for i in range(self.n_joints):
            th = q[..., i] + self.theta_offset[i]
            c, s = np.cos(th), np.sin(th)
            ca, sa = np.cos(self.alpha[i]), np.sin(self.alpha[i])
            A = np.zeros(q.shape[:-1] + (4, 4))
            A[..., 0, 0], A[..., 0, 1], A[..., 0, 2], A[..., 0, 3] = c, -s * ca, s * sa, self.a[i] * c
            A[..., 1, 0], A[..., 1, 1], A[..., 1, 2], A[..., 1, 3] = s, c * ca, -c * sa, self.a[i] * s
            A[..., 2, 1], A[..., 2, 2], A[..., 2, 3] = sa, ca, self.d[i]
            A[..., 3, 3] = 1.0
            T = A if T is None else T @ A
        pose = np.empty(q.shape[:-1] + (6,))
        pose[..., :3] = T[..., :3, 3]
        pose[..., 3] = np.arctan2(T[..., 2, 1], T[..., 2, 2])
        pose[..., 4] = np.arcsin(np.clip(-T[..., 2, 0], -1.0, 1.0))
        pose[..., 5] = np.arctan2(T[..., 1, 0], T[..., 0, 0])
        return pose

    def project(self, T):
        R, t = T[:3, :3], T[:3, 3]
        f = np.empty(2 * self.n_points)
        for i, w in enumerate(self.points):
            pc = R.T @ (w - t)
            f[2 * i] = self.center + self.focal * pc[0] / pc[2]
            f[2 * i + 1] = self.center + self.focal * pc[1] / pc[2]
        return f

    def features(self, q):
        return self.project(self.fkine_all(q)[-1])

    # ---------------------------------------------------------------- device description
    def to_struct(self):
        n, npts = self.n_joints, self.n_points
        if n > _lib.UVS_MAX_N or npts > _lib.UVS_MAX_POINTS:
            raise ValueError('plant exceeds UVS_MAX_N / UVS_MAX_POINTS')
        s = _lib.Plant()
        s.n_joints, s.n_points = n, npts
        for i in range(n):
            s.theta_offset[i], s.d[i], s.a[i] = self.theta_offset[i], self.d[i], self.a[i]
            s.cos_alpha[i], s.sin_alpha[i] = np.cos(self.alpha[i]), np.sin(self.alpha[i])
        for i in range(npts):
            for c in range(3):
                s.points[i][c] = self.points[i][c]
        s.focal, s.center = self.focal, self.center
        s.kind = _lib.PLANT_DH_PINHOLE
        return s


class LinearPlant:
    """Consistent linearised camera f = f0 + J (q - q0) for shapes without a DH model (BASELINE config 5: m = 32, n = 7;
    a 7-joint arm would be redundant for a 6-DoF camera pose and leave the feature Jacobian rank 6).  J, f0, q0 are uploaded
    once and referenced by device pointer from the ``uvs_plant`` struct."""

    def __init__(self, jacobian, f0, q0):
        self.J = np.ascontiguousarray(jacobian, float)
        self.f0, self.q0 = np.ascontiguousarray(f0, float), np.ascontiguousarray(q0, float)
        self._dev = None

    @classmethod
    def random(cls, m=32, n=7, seed=0):
        rng = np.random.default_rng(seed)
        u, _ = np.linalg.qr(rng.normal(size=(m, n)))
        v, _ = np.linalg.qr(rng.normal(size=(n, n)))
        sv = np.geomspace(400.0, 20.0, n)                            # spread like the UR10 interaction matrix (cond ~ 20)
        return cls((u * sv) @ v.T, rng.uniform(90.0, 170.0, m), rng.uniform(-1.0, 1.0, n))

    @property
    def n_joints(self):
        return self.J.shape[1]

    def features(self, q):
        return self.f0 + self.J @ (np.asarray(q, float) - self.q0)

    def to_struct(self, device='cuda'):
        import torch
        if self._dev is None:
            self._dev = [torch.as_tensor(a, device=device) for a in (self.J.ravel(), self.f0, self.q0)]
        s = _lib.Plant()
        s.n_joints, s.n_points, s.kind = self.J.shape[1], self.J.shape[0] // 2, _lib.PLANT_LINEAR
        s.lin_jacobian, s.lin_f0, s.lin_q0 = (t.data_ptr() for t in self._dev)
        return s


class _Clock:
    def __init__(self):
        self.t = 0.0

    def getSimulationTime(self):
        return self.t


class SyntheticRobot:
    """The robot duck-type ``Experiment.run`` expects, backed by a ``SyntheticPlant``: kinematic joints
    (a commanded target is reached in one ``step()``), ``start`` advances the clock once (ur10_simulation.py:57)."""

    def __init__(self, plant=None, dt=0.05, logger=None, visualization=False):
        self.plant = SyntheticPlant.ur10() if plant is None else plant
        self.dt = dt
        self.perspective_angle = self.plant.fov_deg
        self.sim = _Clock()
        self.q = np.zeros(self.plant.n_joints)
        self.q_target = self.q.copy()
        self._Ts = self.plant.fkine_all(self.q)

    def start(self, q=None):
        if q is not None:
            self.q = np.array(q, float)
        self.q_target = self.q.copy()
        self.sim.t = 0.0
        self._Ts = self.plant.fkine_all(self.q)
        self.step()

    def stop(self):
        pass

    def step(self):
        self.q = self.q_target.copy()
        self.sim.t += self.dt

    def getJointsPos(self):
        return self.q

    def setJointsPos(self, q):
        self.q_target = np.array(q, float)

    def fkine(self, recalculate=False, all_transforms=False):
        if recalculate:
            self._Ts = self.plant.fkine_all(self.q)
        return tuple(self._Ts[::-1]) if all_transforms else self._Ts[-1]

    def jacobian(self, recalculate_fkine=False):
        if recalculate_fkine:
            self._Ts = self.plant.fkine_all(self.q)
        return self.plant.jacobian(self._Ts)

    def getCameraRotation(self, recalculate_fkine=False):
        return self.fkine(recalculate_fkine)[:3, :3]

    def getCameraPosition(self, recalculate_fkine=False):
        return self.fkine(recalculate_fkine)[:3, 3]

    def computePose(self, recalculate_fkine=False):
        return camera_pose(self.fkine(True))

    def computeZ(self, n=1, recalculate_fkine=False):
        cam = self.getCameraPosition(recalculate_fkine)
        return np.array([np.linalg.norm(cam - w) for w in self.plant.points[:n]])

    def getCameraImage(self):
        return self, (RESOLUTION, RESOLUTION)

    def features(self):
        return self.plant.project(self.fkine(True))

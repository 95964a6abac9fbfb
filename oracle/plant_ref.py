"""Oracle (test infrastructure): synthetic UR10 + pinhole-camera plant, numpy.

The reference drives a CoppeliaSim scene over ZMQ; the only plant arithmetic
it owns is the DH forward kinematics and geometric Jacobian of
``ur10_simulation.py``.  This module restates those (``fkine`` :97-110,
``jacobian`` :112-139, ``dh`` :204-211, DH table :100-105) and closes the loop
with the consistent pinhole camera documented in SURVEY.md Appendix A:

* camera frame = DH frame 6, ``u = c + F x/z``, ``v = c + F y/z`` with
  ``F = res / (2 tan(fov/2))`` (``experiment.py:97``), res 256, fov 65 deg;
* four point "discs" on the floor that project exactly onto ``desired_f``
  (``config.json:9``) from the straight-down pose ``Q_GOAL``;
* kinematic joints: a commanded target is reached in one ``step()``; ``start``
  steps the clock once (``ur10_simulation.py:57``).

The class is duck-typed to what ``Experiment.run`` calls on a robot
 so the same object drives the unmodified reference in
``gen_golden.py`` *and* the oracle restatement.
"""
import numpy as np

RESOLUTION = 256
FOV_DEG = 65.0
FOCAL = RESOLUTION / (2 * np.tan(0.5 * np.deg2rad(FOV_DEG)))      # experiment.py:97
CENTER = 128.0
DESIRED_F = np.array([149.0, 145.0, 125.0, 121.0, 101.0, 145.0, 125.0, 169.0])   # config.json:9
Q_START = np.array([0.0, 0.0, 1.96349541, 0.0, -1.57079633, 0.0])                # config.json:8
Q_GOAL = np.array([0.0, -np.pi / 8, np.pi / 2 + np.pi / 8, 0.0, -np.pi / 2, 0.0])  # tests/*.py start pose

# (theta offset, d, a, alpha) per joint -- ur10_simulation.py:100-105
DH_TABLE = (
    (0.0, 0.128, 0.0, -np.pi / 2),
    (-np.pi / 2, 0.0, 0.6127, 0.0),
    (0.0, 0.0, 0.5716, 0.0),
    (-np.pi / 2, 0.1639, 0.0, -np.pi / 2),
    (0.0, 0.1157, 0.0, np.pi / 2),
    (np.pi, 0.0922, 0.0, 0.0),
)


def _rot_z_trans_z(theta, d):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s, 0.0, 0.0], [s, c, 0.0, 0.0], [0.0, 0.0, 1.0, d], [0.0, 0.0, 0.0, 1.0]])


def _rot_x_trans_x(alpha, a):
    c, s = np.cos(alpha), np.sin(alpha)
    return np.array([[1.0, 0.0, 0.0, a], [0.0, c, -s, 0.0], [0.0, s, c, 0.0], [0.0, 0.0, 0.0, 1.0]])


def dh_link(theta, d, a, alpha):
    """One DH link transform = Rz(theta)Tz(d) Rx(alpha)Tx(a) (ur10_simulation.py:204-211)."""
    return _rot_z_trans_z(theta, d) @ _rot_x_trans_x(alpha, a)


def fkine_all(q):
    """Cumulative transforms [T_0_1 .. T_0_6] for joint vector q (ur10_simulation.py:97-105)."""
    out = []
    T = np.eye(4)
    for i, (off, d, a, alpha) in enumerate(DH_TABLE):
        link = dh_link(q[i] + off, d, a, alpha)
        T = link if i == 0 else T @ link
        out.append(T)
    return out


def geometric_jacobian(Ts):
    """6x6 geometric Jacobian from the cumulative transforms (ur10_simulation.py:112-139)."""
    p_e = Ts[5][:3, 3]
    cols = []
    z_prev, p_prev = np.array([0.0, 0.0, 1.0]), np.zeros(3)
    for i in range(6):
        cols.append(np.concatenate([np.cross(z_prev, p_e - p_prev), z_prev]))
        z_prev, p_prev = Ts[i][:3, 2], Ts[i][:3, 3]
    return np.stack(cols, axis=1)


def place_discs(desired_f=DESIRED_F, q_goal=Q_GOAL):
    """World positions of the 4 discs so that they project onto desired_f at q_goal (Appendix A)."""
    T = fkine_all(q_goal)[5]
    depth = T[2, 3]
    discs = []
    for i in range(len(desired_f) // 2):
        ray = np.array([(desired_f[2 * i] - CENTER) / FOCAL * depth,
                        (desired_f[2 * i + 1] - CENTER) / FOCAL * depth, depth])
        discs.append(T[:3, 3] + T[:3, :3] @ ray)
    return np.array(discs)


def project(T_cam, discs):
    """Pinhole projection of the discs into the camera at pose T_cam -> f (2*len(discs),)."""
    R, t = T_cam[:3, :3], T_cam[:3, 3]
    f = np.zeros(2 * len(discs))
    for i, d in enumerate(discs):
        pc = R.T @ (d - t)
        f[2 * i] = CENTER + FOCAL * pc[0] / pc[2]
        f[2 * i + 1] = CENTER + FOCAL * pc[1] / pc[2]
    return f


class _Clock:
    def __init__(self):
        self.t = 0.0

    def getSimulationTime(self):
        return self.t


class PinholeUR10:
    """Duck-typed robot backed by this module's kinematics."""

    def __init__(self, dt=0.05, discs=None):
        self.dt = dt
        self.perspective_angle = FOV_DEG
        self.sim = _Clock()
        self.discs = place_discs() if discs is None else np.asarray(discs, float)
        self.q = Q_GOAL.copy()
        self.q_target = self.q.copy()
        self._Ts = fkine_all(self.q)

    # -- lifecycle -----------------------------------------------------------
    def start(self, q):
        self.q = np.array(q, float)
        self.q_target = self.q.copy()
        self.sim.t = 0.0
        self._Ts = fkine_all(self.q)
        self.step()

    def stop(self):
        pass

    def step(self):
        self.q = self.q_target.copy()
        self.sim.t += self.dt

    # -- joints --------------------------------------------------------------
    def getJointsPos(self):
        return self.q

    def setJointsPos(self, q):
        self.q_target = np.array(q, float)

    # -- kinematics ----------------------------------------------------------
    def fkine(self, recalculate=False, all_transforms=False):
        if recalculate:
            self._Ts = fkine_all(self.q)
        if all_transforms:
            return tuple(self._Ts[::-1])
        return self._Ts[5]

    def jacobian(self, recalculate_fkine=False):
        if recalculate_fkine:
            self._Ts = fkine_all(self.q)
        return geometric_jacobian(self._Ts)

    def getCameraRotation(self, recalculate_fkine=False):
        return self.fkine(recalculate_fkine)[:3, :3]

    def getCameraPosition(self, recalculate_fkine=False):
        return self.fkine(recalculate_fkine)[:3, 3]

    def computePose(self, recalculate_fkine=False):
        return np.concatenate([self.fkine(True)[:3, 3], np.zeros(3)])

    def computeZ(self, n=1, recalculate_fkine=False):
        cam = self.getCameraPosition(recalculate_fkine)
        return np.array([np.linalg.norm(cam - d) for d in self.discs[:n]])

    # -- perception ----------------------------------------------------------
    def getCameraImage(self):
        return self, (RESOLUTION, RESOLUTION)

    def features(self):
        return project(self.fkine(True), self.discs)


# ---------------------------------------------------------------------------------------------------------------------
# Synthetic camera frames for the circle-detector fixtures: filled discs on a grey background, RGB uint8, row 0 at the
# bottom like the simulator's vision sensor (the detectors flip the frame first, utils.py:13).
DISC_COLOURS = {'red': (255, 0, 0), 'green': (0, 255, 0), 'blue': (0, 0, 255), 'pink': (255, 0, 255)}


def render_discs(centres, radii, size=256, background=96, soften=False):
    """centres: {colour: (u, v)} in pixels of the *flipped* frame (u right, v down); returns (size, size, 3) uint8.
    soften=True adds an anti-aliased rim (values 97..254) that the 250 threshold must reject."""
    vv, uu = np.mgrid[0:size, 0:size].astype(float)
    img = np.full((size, size, 3), background, np.uint8)
    for colour, (u, v) in centres.items():
        d = np.hypot(uu - u, vv - v)
        r = radii[colour] if isinstance(radii, dict) else radii
        inside = d <= r
        img[inside] = DISC_COLOURS[colour]
        if soften:
            rim = (d > r) & (d <= r + 1.5)
            a = ((r + 1.5 - d[rim]) / 1.5)[:, None]
            img[rim] = (a * np.array(DISC_COLOURS[colour]) + (1 - a) * background).astype(np.uint8)
    return img[::-1].copy()                                  # hand over unflipped, as the sensor does

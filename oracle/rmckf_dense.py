"""Oracle (test infrastructure): dense numpy restatement of the reference servo loop.

Mirrors ``Experiment.run`` of the reference (experiment.py:48-359) operation for
operation -- dense ``mn x mn`` covariance, ``np.kron`` regressor, LAPACK
``inv``/``pinv`` -- so that it reproduces the reference to rounding.  This is
the form that is pinned against ``tests/golden/closed_*.npz`` (generated from
the unmodified reference by ``oracle/gen_golden.py``) and the form timed as the
CPU baseline.  ``oracle/rmckf_block.py`` is the per-row restatement for
arbitrary (m, n).

Method names follow the reference enum (experiment.py:6-11); the paper's RMCKF
is ``GMCKF``.
"""
import numpy as np

KF, MCKF, IMCCKF, GMCKF = 'KF', 'MCKF', 'IMCCKF', 'GMCKF'
SUCCESS, FAIL = 0, 1


def gaussian_kernel(e, bw):
    """utils.py:171-172."""
    return np.exp(-0.5 * e ** 2 / bw ** 2)


def analytic_initial_guess(robot, f, m, n=6):
    """Interaction-matrix initial guess X0 = vec(J_img kron(I2, R^T) J_robot) (experiment.py:94-114)."""
    depth = robot.computeZ(m // 2)
    focal = 256 / (2 * np.tan(0.5 * np.deg2rad(robot.perspective_angle)))
    Ji = np.zeros((m, n))
    for i in range(m // 2):
        u, v = f[2 * i], f[2 * i + 1]
        Ji[2 * i] = [-focal / depth[i], 0.0, u / depth[i], u * v / focal, -(focal ** 2 + u ** 2) / focal, v]
        Ji[2 * i + 1] = [0.0, -focal / depth[i], v / depth[i], (focal ** 2 + v ** 2) / focal, -u * v / focal, -u]
    Jf = Ji @ np.kron(np.eye(2), robot.getCameraRotation().T) @ robot.jacobian()
    return Jf.reshape((m * n, 1))


def bandwidth(kernel_bw, annealing, k, k_max):
    """sigma_k (experiment.py:267-271)."""
    return kernel_bw + 100 * (1 - k / k_max) if annealing else kernel_bw


class DenseFilter:
    """State (X, P) of one estimator and its per-step update (experiment.py:70-76, 164-298)."""

    def __init__(self, m, n, method=GMCKF, kernel_bw=10.0, annealing=False, k_max=300,
                 fpi_threshold=0.1, fpi_epoch_max=1000, x0=None):
        self.m, self.n, self.method = m, n, method
        self.kernel_bw, self.annealing, self.k_max = kernel_bw, annealing, k_max
        self.fpi_threshold, self.fpi_epoch_max = fpi_threshold, fpi_epoch_max
        mn = m * n
        self.X = np.zeros((mn, 1)) if x0 is None else np.array(x0, float).reshape(mn, 1)
        self.P = np.eye(mn)
        self.Q = np.eye(mn)
        self.R = np.eye(m)
        self.H = np.zeros((m, mn))
        self.K = np.zeros((mn, m))
        self.Br = np.linalg.cholesky(self.R)
        self.Br_inv = np.linalg.inv(self.Br)
        self.first = True
        self.fpi_iterations, self.fpi_skipped = 0, False
        self.kappa = np.ones(m)
        self.sigma = -1.0

    def step(self, Z, dq_prev, k):
        """Predict + correct with measurement Z (m,) and regressor dq_prev (n,). Returns kappa (m,)."""
        m, n, mn = self.m, self.n, self.m * self.n
        X, P, R = self.X, self.P, self.R
        Z = np.asarray(Z, float).reshape(m, 1)
        P = P + self.Q                                                     # :167
        if self.first:                                                     # :183-188
            self.first = False
        else:
            self.H = np.kron(np.eye(m), np.asarray(dq_prev, float).ravel())
        H, K = self.H, self.K
        skip = False
        kappa = np.ones(m)
        if self.method == KF:                                              # :191-193
            K = P @ H.T @ np.linalg.inv(H @ P @ H.T + R)
            X = X + K @ (Z - H @ X)
        elif self.method == MCKF:                                          # :194-250
            bw = self.sigma = bandwidth(self.kernel_bw, self.annealing, k, self.k_max)
            Bp = np.linalg.cholesky(P)
            Br = np.linalg.cholesky(R)
            B = np.block([[Bp, np.zeros((mn, m))], [np.zeros((m, mn)), Br]])
            B_inv = np.linalg.pinv(B)
            D = B_inv @ np.vstack([X, Z])
            W = B_inv @ np.vstack([np.eye(mn), H])
            Xc, diff, it = X.copy(), np.inf, 0
            while diff > self.fpi_threshold and it < self.fpi_epoch_max:
                e = D - W @ Xc
                Cx = np.diag([gaussian_kernel(e[i, 0], bw) for i in range(mn)])
                Cy = np.diag([gaussian_kernel(e[i, 0], bw) for i in range(mn, mn + m)])
                P_hat = Bp @ np.linalg.inv(Cx) @ Bp.T
                try:
                    R_hat = Br @ np.linalg.inv(Cy) @ Br.T
                except np.linalg.LinAlgError:
                    skip = True
                    break
                K = P_hat @ H.T @ np.linalg.inv(H @ P_hat @ H.T + R_hat)
                Xc_old = Xc.copy()
                Xc = X + K @ (Z - H @ X)
                diff = np.linalg.norm(Xc - Xc_old) / np.linalg.norm(Xc_old)
                it += 1
                if it == self.fpi_epoch_max:
                    skip = True
            self.fpi_iterations, self.fpi_skipped = it, skip
            if not skip:
                X = Xc
        elif self.method == IMCCKF:                                        # :251-265
            bw = self.sigma = bandwidth(self.kernel_bw, self.annealing, k, self.k_max)
            innov = Z - H @ X
            nrm = np.sqrt(innov.T @ np.linalg.inv(R) @ innov)
            Cy = gaussian_kernel(nrm, bw)
            R_e = Cy * H @ P @ H.T + R
            K = Cy * P @ H.T @ np.linalg.inv(R_e)
            X = X + K @ innov
        elif self.method == GMCKF:                                         # :266-294
            bw = self.sigma = bandwidth(self.kernel_bw, self.annealing, k, self.k_max)
            e = self.Br_inv @ Z - self.Br_inv @ H @ X
            Cy = np.diag([gaussian_kernel(e[i, 0], bw) for i in range(m)])
            try:
                Cy_inv = np.linalg.inv(Cy + 0.001 ** 2 * np.eye(m))
                R_hat = self.Br @ Cy_inv @ self.Br.T
                S = H @ P @ H.T + R_hat
                K = P @ H.T @ np.linalg.inv(S)
                X = X + K @ (Z - H @ X)
            except np.linalg.LinAlgError:
                skip = True
            kappa = np.array([gaussian_kernel(e[i, 0], bw) for i in range(m)])   # :308
        else:
            raise ValueError(self.method)
        if not skip:                                                       # :296-297 Joseph form
            IKH = np.eye(mn) - K @ H
            P = IKH @ P @ IKH.T + K @ R @ K.T
        self.X, self.P, self.K, self.kappa = X, P, K, kappa
        return kappa


def control_law(X, m, n, err, kappa, gain):
    """dq = -gain * pinv(J) (kappa o err)  (experiment.py:300-312). Raises LinAlgError on non-finite J."""
    J = X.reshape((m, n))
    return -gain * np.linalg.pinv(J) @ (kappa.reshape((m, 1)) * err.reshape((m, 1)))


def run_closed_loop(robot, q_start, desired_f, noise_next, t_s, t_max, gain, method=GMCKF,
                    initial_guess=True, kernel_bw=10.0, annealing=False, fpi_threshold=0.1,
                    fpi_epoch_max=1000, x0=None, capture=False):
    """One servo trial (experiment.py:48-359).  ``noise_next()`` returns the next (m,) noise sample
    (or ``None`` for a noise-free run).  Returns a dict of the reference's logs plus, when
    ``capture``, the per-step X / block-diagonal of P / commanded dq."""
    desired_f = np.asarray(desired_f, float)
    m, n = len(desired_f), 6
    k_max = int(t_max / t_s)
    robot.start(q_start)
    f = np.zeros(m)
    noise = np.zeros(m)
    if initial_guess:
        f = robot.getCameraImage()[0].features()
        x0 = analytic_initial_guess(robot, f, m, n)
    elif x0 is None:
        raise ValueError('x0 required when initial_guess is False (reference draws it unseeded, experiment.py:117)')
    filt = DenseFilter(m, n, method, kernel_bw, annealing, k_max, fpi_threshold, fpi_epoch_max, x0)
    dq = np.zeros((n, 1))
    logs = {key: [] for key in ('t', 'err', 'q', 'f', 'noise', 'cam', 'X', 'Pblk', 'dq', 'sigma', 'kappa', 'fpi_iterations', 'fpi_skipped')}
    status, k = SUCCESS, 0
    while (t := robot.sim.getSimulationTime()) < t_max:
        f_old = f.copy()
        f = robot.getCameraImage()[0].features()
        if noise_next is not None:
            noise = noise_next()
            f = f + noise
        kappa = filt.step(f - f_old, dq, k)
        err = f - desired_f
        if capture and method == MCKF:                                     # recorded for the FAILing step too, like the tracer at experiment.py:302
            logs['fpi_iterations'].append(filt.fpi_iterations); logs['fpi_skipped'].append(filt.fpi_skipped)
        try:
            dq = control_law(filt.X, m, n, err, kappa, gain)
        except np.linalg.LinAlgError:
            status = FAIL
            break
        q_now = robot.getJointsPos().copy()
        logs['t'].append(t); logs['err'].append(err); logs['q'].append(q_now); logs['f'].append(f.copy())
        logs['noise'].append(np.array(noise, copy=True)); logs['cam'].append(robot.computePose())
        logs['sigma'].append(filt.sigma)
        if capture:
            logs['X'].append(filt.X.ravel().copy()); logs['dq'].append(dq.ravel().copy()); logs['kappa'].append(kappa.copy())
            logs['Pblk'].append(np.stack([filt.P[i * n:(i + 1) * n, i * n:(i + 1) * n] for i in range(m)]))
        k += 1
        robot.setJointsPos(q_now + dq.ravel() * t_s)
        robot.step()
    robot.stop()
    out = {key: np.array(val) for key, val in logs.items() if len(val)}
    out.update(status=status, k_done=k, P_final=filt.P.copy(), X_final=filt.X.ravel().copy())
    return out


def run_replay(f_seq, dq_seq, x0, desired_f, gain, method=GMCKF, kernel_bw=10.0, annealing=False,
               k_max=300, fpi_threshold=0.1, fpi_epoch_max=1000):
    """Open-loop replay: f_seq (K+1, m) observed features (row 0 = f_old of the first step), dq_seq (K, n)
    the regressor used at each step (row 0 is ignored: H = 0 on the first iteration, experiment.py:183).
    Returns per-step X (K, mn), err (K, m), kappa (K, m), commanded dq (K, n)."""
    f_seq, dq_seq = np.asarray(f_seq, float), np.asarray(dq_seq, float)
    K, m, n = len(dq_seq), f_seq.shape[1], dq_seq.shape[1]
    filt = DenseFilter(m, n, method, kernel_bw, annealing, k_max, fpi_threshold, fpi_epoch_max, x0)
    Xs, errs, kaps, cmds = [], [], [], []
    for k in range(K):
        kappa = filt.step(f_seq[k + 1] - f_seq[k], dq_seq[k], k)
        err = f_seq[k + 1] - np.asarray(desired_f, float)
        cmds.append(control_law(filt.X, m, n, err, kappa, gain).ravel())
        Xs.append(filt.X.ravel().copy()); errs.append(err); kaps.append(kappa.copy())
    return dict(X=np.array(Xs), err=np.array(errs), kappa=np.array(kaps), dq_cmd=np.array(cmds), P_final=filt.P.copy())


def trial_stats(err, t):
    """Per-trial ||ISE||_2, ||IAE||_2, ||ITAE||_2 over the features (results/plot_errorbar.m:39-84)."""
    err, t = np.asarray(err, float), np.asarray(t, float)
    ise = np.sum(err * err, axis=0)
    iae = np.sum(np.abs(err), axis=0)
    itae = t @ np.abs(err)
    return np.array([np.linalg.norm(ise), np.linalg.norm(iae), np.linalg.norm(itae)])

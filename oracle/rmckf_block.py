"""Oracle (test infrastructure): per-row ("block") restatement of the estimators for any (m, n).

Because ``H = kron(I_m, dq^T)`` (experiment.py:188) and ``P0 = Q = I``, ``R = I``
(experiment.py:73-76), the reference's dense ``mn x mn`` covariance is exactly
block diagonal: m independent n x n blocks that share the regressor ``h = dq``
(SURVEY.md fact 4).  This module evaluates the same recursion row by row:

    P_i += I;  nu_i = Z_i - x_i.h;  c_i = exp(-nu_i^2 / (2 sigma^2));  r_i = 1/(c_i + 1e-6)
    g_i = P_i h;  s_i = h.g_i + r_i;  k_i = g_i / s_i;  x_i += k_i nu_i
    P_i = (I - k_i h^T) P_i (I - k_i h^T)^T + k_i k_i^T          (Joseph, R = 1)

It is validated against ``rmckf_dense`` (and through it against the reference
fixtures) at (m, n) = (8, 6) and is the oracle for shapes the reference cannot
run ((2, 6), (32, 7), ...; experiment.py:54,170-177 hard-wire m = 8, n = 6).
"""
import numpy as np

KF, MCKF, IMCCKF, GMCKF = 'KF', 'MCKF', 'IMCCKF', 'GMCKF'
REG = 0.001 ** 2                                                    # experiment.py:280


class BlockFilter:
    def __init__(self, m, n, x0=None, method=GMCKF, kernel_bw=10.0, annealing=False, k_max=300, fpi_threshold=0.1, fpi_epoch_max=1000):
        self.m, self.n, self.method = m, n, method
        self.kernel_bw, self.annealing, self.k_max = kernel_bw, annealing, k_max
        self.fpi_threshold, self.fpi_epoch_max = fpi_threshold, fpi_epoch_max
        self.fpi_iterations = 0
        self.X = np.zeros((m, n)) if x0 is None else np.array(x0, float).reshape(m, n)
        self.P = np.tile(np.eye(n), (m, 1, 1))
        self.first = True
        self.sigma = -1.0

    def step(self, Z, dq_prev, k):
        m, n = self.m, self.n
        Z = np.asarray(Z, float).ravel()
        h = np.zeros(n) if self.first else np.asarray(dq_prev, float).ravel()
        self.first = False
        P = self.P + np.eye(n)                                      # predict
        if self.method == MCKF:
            return self._step_mckf(P, Z, h, k)
        nu = Z - self.X @ h                                         # innovation, row by row
        g = P @ h                                                   # (m, n)
        a = g @ h                                                   # h^T P_i h
        kappa = np.ones(m)
        if self.method == KF:
            gain_scale, r = np.ones(m), np.ones(m)
        elif self.method == IMCCKF:                                 # one scalar weight couples the rows
            self.sigma = self._sigma(k)
            c = np.exp(-0.5 * np.sqrt(nu @ nu) ** 2 / self.sigma ** 2)
            gain_scale, r = np.full(m, c), np.ones(m)               # K = c P H^T (c H P H^T + R)^-1
        elif self.method == GMCKF:
            self.sigma = self._sigma(k)
            kappa = np.exp(-0.5 * nu ** 2 / self.sigma ** 2)
            gain_scale, r = np.ones(m), 1.0 / (kappa + REG)
        else:
            raise ValueError(self.method)
        s = gain_scale * a + r
        kk = (gain_scale / s)[:, None] * g                          # (m, n) gain rows
        self.X = self.X + kk * nu[:, None]
        A = np.eye(n)[None] - kk[:, :, None] * h[None, None, :]     # I - k h^T
        self.P = A @ P @ np.transpose(A, (0, 2, 1)) + kk[:, :, None] * kk[:, None, :]
        return kappa

    def _step_mckf(self, P, Z, h, k):
        """Fixed-point MCKF (experiment.py:194-250) in block form.  With B = blkdiag(chol(P), chol(R)) block diagonal,
        D - W X_c = [L_i^-1 (x_i - xc_i); z_i - h.xc_i]; P_hat_i = L_i Cx_i^-1 L_i^T, R_hat_i = 1/Cy_i; the stop test uses the
        norm over all rows (:244) and a zero in Cy (8x8 matrix, :231) or max epoch (:246-248) skips the whole correction."""
        m, n = self.m, self.n
        self.sigma = bw = self._sigma(k)
        Lc = np.linalg.cholesky(P)                                  # (m, n, n) lower factors
        X = self.X
        nu0 = Z - X @ h                                             # Z - H X: the gain is applied to the *prior* innovation (:242)
        Xc, diff, it, skip = X.copy(), np.inf, 0, False
        kk = np.zeros((m, n))
        while diff > self.fpi_threshold and it < self.fpi_epoch_max:
            ex = np.linalg.solve(Lc, (X - Xc)[:, :, None])[:, :, 0]    # L_i^-1 (x_i - xc_i)
            ez = Z - Xc @ h
            cx = np.exp(-0.5 * ex ** 2 / bw ** 2)
            cy = np.exp(-0.5 * ez ** 2 / bw ** 2)
            if np.any(cy == 0.0):                                   # inv(Cy) raises LinAlgError (:231-236)
                skip = True
                break
            with np.errstate(divide='ignore', over='ignore', invalid='ignore'):
                P_hat = Lc @ (np.transpose(Lc, (0, 2, 1)) / cx[:, :, None])
                g = P_hat @ h
                kk = g / ((g @ h) + 1.0 / cy)[:, None]
                # A weight that is subnormal but not 0 (its reciprocal overflows): inv() returns inf without raising, and the DENSE products
                # of the reference, Br @ inv(Cy) @ Br.T (:232) resp. Bp @ inv(Cx) @ Bp.T (:223), turn 0 * inf into NaN -- the whole gain, the
                # state and with it the trial (pinv raises in the control law, :312-316).  A Cx of exactly 0 makes inv(Cx) raise outside any
                # try (:223, the reference's sweep aborts); it is folded into the same FAIL here (policies).
                if np.any(np.isinf(1.0 / cy)) or np.any(np.isinf(1.0 / cx)):
                    kk = np.full((m, n), np.nan)
            Xc_old = Xc
            Xc = X + kk * nu0[:, None]
            diff = np.linalg.norm(Xc - Xc_old) / np.linalg.norm(Xc_old)
            it += 1
            if it == self.fpi_epoch_max:
                skip = True
        self.fpi_iterations = it
        if not skip:
            self.X = Xc
            A = np.eye(n)[None] - kk[:, :, None] * h[None, None, :]
            self.P = A @ P @ np.transpose(A, (0, 2, 1)) + kk[:, :, None] * kk[:, None, :]
        else:
            self.P = P
        return np.ones(m)

    def _sigma(self, k):
        return self.kernel_bw + 100 * (1 - k / self.k_max) if self.annealing else self.kernel_bw


def control_law(X, err, kappa, gain):
    """dq = -gain pinv(X) (kappa o err); works for m >= n (least squares) and m < n (minimum norm)."""
    return -gain * (np.linalg.pinv(X) @ (kappa * err))


def run_replay(f_seq, dq_seq, x0, desired_f, gain, method=GMCKF, kernel_bw=10.0, annealing=False, k_max=300, fpi_threshold=0.1,
               fpi_epoch_max=1000):
    f_seq, dq_seq = np.asarray(f_seq, float), np.asarray(dq_seq, float)
    K, m, n = len(dq_seq), f_seq.shape[1], dq_seq.shape[1]
    filt = BlockFilter(m, n, x0, method, kernel_bw, annealing, k_max, fpi_threshold, fpi_epoch_max)
    Xs, errs, kaps, cmds, its = [], [], [], [], []
    for k in range(K):
        kappa = filt.step(f_seq[k + 1] - f_seq[k], dq_seq[k], k)
        err = f_seq[k + 1] - np.asarray(desired_f, float)
        cmds.append(control_law(filt.X, err, kappa, gain))
        Xs.append(filt.X.ravel().copy()); errs.append(err); kaps.append(kappa.copy()); its.append(filt.fpi_iterations)
    return dict(X=np.array(Xs), err=np.array(errs), kappa=np.array(kaps), dq_cmd=np.array(cmds), P_final=filt.P.copy(), fpi_iterations=np.array(its))


def run_closed_loop(plant, q_start, desired_f, noise_seq, t_s, t_max, gain, x0, method=GMCKF,
                    kernel_bw=10.0, annealing=False, initial_guess=True, fpi_threshold=0.1, fpi_epoch_max=1000):
    """Closed loop on a functional plant: ``plant(q) -> f`` (noise-free features at joints q).
    ``noise_seq`` (K, m) or None.  ``initial_guess`` selects what the first ``f_old`` is: the noise-free features seen while
    forming the analytic X0 (experiment.py:90) or zeros when X0 is supplied (experiment.py:56).
    Returns err (k, m), q (k, n), X (k, mn), t (k,), status, k_done."""
    desired_f = np.asarray(desired_f, float)
    m, n = len(desired_f), len(q_start)
    k_max = int(t_max / t_s)
    filt = BlockFilter(m, n, x0, method, kernel_bw, annealing, k_max, fpi_threshold, fpi_epoch_max)
    q = np.array(q_start, float)
    f = plant(q) if initial_guess else np.zeros(m)
    dq = np.zeros(n)
    t, k = t_s, 0                                                   # start() steps the clock once
    ts, errs, qs, Xs, its = [], [], [], [], []
    status = 0
    while t < t_max:
        f_old = f
        f = plant(q) + (noise_seq[k] if noise_seq is not None else 0.0)
        kappa = filt.step(f - f_old, dq, k)
        err = f - desired_f
        if not np.all(np.isfinite(filt.X)):                         # pinv raises -> FAIL (experiment.py:313-316)
            status = 1
            break
        dq = control_law(filt.X, err, kappa, gain)
        ts.append(t); errs.append(err); qs.append(q.copy()); Xs.append(filt.X.ravel().copy()); its.append(filt.fpi_iterations)
        k += 1
        q = q + dq * t_s
        t += t_s
    return dict(t=np.array(ts), err=np.array(errs), q=np.array(qs), X=np.array(Xs), status=status, k_done=k,
                P_final=filt.P.copy(), fpi_iterations=np.array(its))

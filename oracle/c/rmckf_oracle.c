/*
 * ORACLE (test infrastructure only -- never linked into or called by the product library).
 *
 * Plain-C restatement of the reference's servo loop in per-row ("block") form, for bulk checks of the HIP path at sizes
 * where the numpy oracle is too slow, and as an optional CPU timing reference.  Follows, line by line:
 *   experiment.py:125-343  loop (measurement, estimator, control law, actuation)
 *   experiment.py:166-167  predict            :170-188 measurement / regressor
 *   experiment.py:191-193  KF                  :251-265 IMCCKF            :266-294 GMCKF (the paper's RMCKF)
 *   experiment.py:194-250  MCKF: fixed-point iteration per step (Cholesky factor of every predicted block, weights Cx / Cy, stop test
 *                          over all rows, a zero Cy or the epoch cap skips the correction; a weight whose reciprocal overflows
 *                          turns the dense products of the reference into NaN and the trial FAILs)
 *   experiment.py:296-297  Joseph covariance update (evaluated as written: A P A^T + k k^T per block)
 *   experiment.py:300-316  control law with numpy.linalg.pinv semantics (SVD, singular values <= 1e-15 * max dropped;
 *                          a non-finite Jacobian makes pinv raise -> FAIL)
 *   experiment.py:86-114   analytic initial guess        utils.py:171-172 gaussianKernel
 *   ur10_simulation.py:97-139,204-211  DH forward kinematics and geometric Jacobian; pinhole camera of SURVEY.md Appendix A
 * It is validated against tests/golden/closed_*.npz (outputs of the unmodified reference) by tests/test_oracle_c.py.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MAXM 32
#define MAXN 8
#define MAXPTS 16

typedef struct {
    int32_t m, n, method, annealing, k_max, steps, initial_guess, pad;
    double kernel_bw, anneal_span, gain, dt, reg;
    double desired[MAXM];
    double fpi_threshold;
    int32_t fpi_epoch_max, pad2;
} oracle_params;

typedef struct {
    int32_t n_joints, n_points;
    double theta_offset[MAXN], d[MAXN], a[MAXN], alpha[MAXN];
    double points[MAXPTS][3];
    double focal, center;
    /* linear consistent plant f = lin_f0 + lin_J (q - lin_q0) (BASELINE config 5: no DH model at m = 32, n = 7); used when lin_J != NULL */
    const double *lin_J, *lin_f0, *lin_q0;
} oracle_plant;

static double gaussian_kernel(double e, double bw) { return exp(-0.5 * (e * e) / (bw * bw)); }

/* T_0_i for i = 1..n (row-major 4x4), ur10_simulation.py:97-110 */
static void fkine_all(const oracle_plant *pl, const double *q, double T[][16]) {
    double cur[16];
    for (int i = 0; i < pl->n_joints; ++i) {
        const double th = q[i] + pl->theta_offset[i], c = cos(th), s = sin(th), ca = cos(pl->alpha[i]), sa = sin(pl->alpha[i]);
        const double A[16] = {c, -s * ca, s * sa, pl->a[i] * c, s, c * ca, -c * sa, pl->a[i] * s, 0, sa, ca, pl->d[i], 0, 0, 0, 1};
        if (i == 0) {
            memcpy(cur, A, sizeof cur);
        } else {
            double nxt[16];
            for (int r = 0; r < 4; ++r)
                for (int cc = 0; cc < 4; ++cc) {
                    double acc = 0;
                    for (int k = 0; k < 4; ++k) acc += cur[4 * r + k] * A[4 * k + cc];
                    nxt[4 * r + cc] = acc;
                }
            memcpy(cur, nxt, sizeof cur);
        }
        memcpy(T[i], cur, sizeof cur);
    }
}

static void features(const oracle_plant *pl, const double *Tc, double *f) {
    for (int i = 0; i < pl->n_points; ++i) {
        double dw[3], pc[3];
        for (int r = 0; r < 3; ++r) dw[r] = pl->points[i][r] - Tc[4 * r + 3];
        for (int c = 0; c < 3; ++c) pc[c] = Tc[c] * dw[0] + Tc[4 + c] * dw[1] + Tc[8 + c] * dw[2];       /* R^T (w - t) */
        f[2 * i] = pl->center + pl->focal * pc[0] / pc[2];
        f[2 * i + 1] = pl->center + pl->focal * pc[1] / pc[2];
    }
}

/* numpy.linalg.pinv(J) @ y through a one-sided Jacobi SVD of the taller of J / J^T.  Returns 0, or -1 when J is non-finite. */
static int pinv_apply(const double *J, int m, int n, const double *y, double *out) {
    for (int i = 0; i < m * n; ++i)
        if (!isfinite(J[i])) return -1;
    const int tall = m >= n, rows = tall ? m : n, cols = tall ? n : m;
    double A[MAXM * MAXM], V[MAXM * MAXM];                       /* A: rows x cols working copy, V: cols x cols */
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) A[r * cols + c] = tall ? J[r * n + c] : J[c * n + r];
    for (int i = 0; i < cols * cols; ++i) V[i] = 0;
    for (int i = 0; i < cols; ++i) V[i * cols + i] = 1;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < cols - 1; ++p)
            for (int q2 = p + 1; q2 < cols; ++q2) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < rows; ++r) {
                    al += A[r * cols + p] * A[r * cols + p];
                    be += A[r * cols + q2] * A[r * cols + q2];
                    ga += A[r * cols + p] * A[r * cols + q2];
                }
                if (ga == 0 || fabs(ga) <= 1e-300) continue;
                const double rel = fabs(ga) / sqrt(al * be);
                if (rel > off) off = rel;
                if (rel < 1e-17) continue;
                const double zeta = (be - al) / (2 * ga), t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                const double c = 1 / sqrt(1 + t * t), s = c * t;
                for (int r = 0; r < rows; ++r) {
                    const double ap = A[r * cols + p], aq = A[r * cols + q2];
                    A[r * cols + p] = c * ap - s * aq;
                    A[r * cols + q2] = s * ap + c * aq;
                }
                for (int r = 0; r < cols; ++r) {
                    const double vp = V[r * cols + p], vq = V[r * cols + q2];
                    V[r * cols + p] = c * vp - s * vq;
                    V[r * cols + q2] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double sv[MAXM], smax = 0;
    for (int c = 0; c < cols; ++c) {
        double s2 = 0;
        for (int r = 0; r < rows; ++r) s2 += A[r * cols + c] * A[r * cols + c];
        sv[c] = sqrt(s2);
        if (sv[c] > smax) smax = sv[c];
    }
    const double cutoff = 1e-15 * smax;                          /* numpy pinv default rcond */
    /* J = U S V^T (tall) or J^T = U S V^T (wide), with U S = A.  pinv(J) y = V S^-1 U^T y  resp.  U S^-1 V^T y. */
    for (int i = 0; i < n; ++i) out[i] = 0;
    for (int c = 0; c < cols; ++c) {
        if (!(sv[c] > cutoff)) continue;
        if (tall) {
            double uy = 0;
            for (int r = 0; r < rows; ++r) uy += A[r * cols + c] * y[r];
            uy /= sv[c] * sv[c];
            for (int i = 0; i < n; ++i) out[i] += V[i * cols + c] * uy;
        } else {
            double vy = 0;
            for (int r = 0; r < cols; ++r) vy += V[r * cols + c] * y[r];
            vy /= sv[c] * sv[c];
            for (int i = 0; i < n; ++i) out[i] += A[i * cols + c] * vy;
        }
    }
    return 0;
}

static void initial_guess(const oracle_plant *pl, const double *q, int m, int n, double *X, double *f) {
    double T[MAXN][16];
    fkine_all(pl, q, T);
    const double *Tc = T[n - 1];
    features(pl, Tc, f);
    double Jr[6][MAXN];                                          /* geometric Jacobian, ur10_simulation.py:112-139 */
    for (int i = 0; i < n; ++i) {
        double z[3] = {0, 0, 1}, o[3] = {0, 0, 0};
        if (i > 0)
            for (int r = 0; r < 3; ++r) { z[r] = T[i - 1][4 * r + 2]; o[r] = T[i - 1][4 * r + 3]; }
        const double dx = Tc[3] - o[0], dy = Tc[7] - o[1], dz = Tc[11] - o[2];
        Jr[0][i] = z[1] * dz - z[2] * dy; Jr[1][i] = z[2] * dx - z[0] * dz; Jr[2][i] = z[0] * dy - z[1] * dx;
        Jr[3][i] = z[0]; Jr[4][i] = z[1]; Jr[5][i] = z[2];
    }
    double Jc[6][MAXN];                                          /* kron(I2, R^T) J */
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r) {
            Jc[r][i] = Tc[r] * Jr[0][i] + Tc[4 + r] * Jr[1][i] + Tc[8 + r] * Jr[2][i];
            Jc[3 + r][i] = Tc[r] * Jr[3][i] + Tc[4 + r] * Jr[4][i] + Tc[8 + r] * Jr[5][i];
        }
    const double F = pl->focal;
    for (int pt = 0; pt < m / 2; ++pt) {
        const double u = f[2 * pt], v = f[2 * pt + 1];
        double dd = 0;
        for (int r = 0; r < 3; ++r) dd += (Tc[4 * r + 3] - pl->points[pt][r]) * (Tc[4 * r + 3] - pl->points[pt][r]);
        const double Z = sqrt(dd);
        const double ru[6] = {-F / Z, 0, u / Z, u * v / F, -(F * F + u * u) / F, v};
        const double rv[6] = {0, -F / Z, v / Z, (F * F + v * v) / F, -u * v / F, -u};
        for (int j = 0; j < n; ++j) {
            double au = 0, av = 0;
            for (int k = 0; k < 6; ++k) { au += ru[k] * Jc[k][j]; av += rv[k] * Jc[k][j]; }
            X[(2 * pt) * n + j] = au;
            X[(2 * pt + 1) * n + j] = av;
        }
    }
}

/* Fixed-point MCKF correction of one step (experiment.py:194-250) on the predicted blocks P (already P + Q).  Z: measurement, h: regressor.
 * Updates X and P in place; returns the reference's `epoch` counter (passes completed; 0 when the first pass met a zero weight). */
static int mckf_step(int m, int n, double *X, double P[][MAXN * MAXN], const double *Z, const double *h, double bw, double thr, int cap) {
    static const double kInf = 1.0 / 0.0;
    double L[MAXM][MAXN * MAXN], nu0[MAXM], Xc[MAXM * MAXN], kk[MAXM * MAXN];
    for (int i = 0; i < m; ++i) {                                /* Bp = cholesky(P), block by block (:203) */
        for (int j = 0; j < n; ++j)
            for (int r = 0; r < n; ++r) L[i][r * n + j] = 0;
        for (int j = 0; j < n; ++j) {
            double d = P[i][j * n + j];
            for (int c = 0; c < j; ++c) d -= L[i][j * n + c] * L[i][j * n + c];
            const double ljj = sqrt(d);
            L[i][j * n + j] = ljj;
            for (int r = j + 1; r < n; ++r) {
                double v = P[i][r * n + j];
                for (int c = 0; c < j; ++c) v -= L[i][r * n + c] * L[i][j * n + c];
                L[i][r * n + j] = v / ljj;
            }
        }
        nu0[i] = Z[i];
        for (int j = 0; j < n; ++j) nu0[i] -= X[i * n + j] * h[j];    /* Z - H X: the gain is applied to the prior innovation (:242) */
    }
    memcpy(Xc, X, sizeof(double) * m * n);
    for (int i = 0; i < m * n; ++i) kk[i] = 0;
    double diff = kInf;
    int it = 0, skip = 0;
    while (diff > thr && it < cap) {                             /* :213 */
        double cx[MAXM][MAXN], cy[MAXM];
        int zero = 0, poison = 0;
        for (int i = 0; i < m; ++i) {
            double ex[MAXN], ez = Z[i];
            for (int r = 0; r < n; ++r) {                        /* L ex = x - xc (rows of D - W Xc, :215) */
                double v = X[i * n + r] - Xc[i * n + r];
                for (int c = 0; c < r; ++c) v -= L[i][r * n + c] * ex[c];
                ex[r] = v / L[i][r * n + r];
                cx[i][r] = gaussian_kernel(ex[r], bw);
                if (isinf(1.0 / cx[i][r])) poison = 1;          /* inv(Cx): inf entries (or, for an exact 0, an uncaught LinAlgError) */
            }
            for (int j = 0; j < n; ++j) ez -= Xc[i * n + j] * h[j];
            cy[i] = gaussian_kernel(ez, bw);
            if (cy[i] == 0.0) zero = 1;
            else if (isinf(1.0 / cy[i])) poison = 1;
        }
        if (zero) { skip = 1; break; }                           /* inv(Cy) raises: skip the correction (:225-236) */
        double num = 0, den = 0;
        for (int i = 0; i < m; ++i) {
            double t[MAXN], g[MAXN], a = 0;
            for (int j = 0; j < n; ++j) {                        /* t = Cx^-1 L^T h */
                double v = 0;
                for (int r = j; r < n; ++r) v += L[i][r * n + j] * h[r];
                t[j] = v / cx[i][j];
            }
            for (int r = 0; r < n; ++r) {                        /* g = L t = P_hat h */
                double v = 0;
                for (int c = 0; c <= r; ++c) v += L[i][r * n + c] * t[c];
                g[r] = v;
                a += h[r] * v;
            }
            const double s = a + 1.0 / cy[i];
            for (int l = 0; l < n; ++l) {
                /* 0 * inf of the reference's dense Br Cy^-1 Br^T / Bp Cx^-1 Bp^T poisons the whole gain (:223, :232) */
                kk[i * n + l] = poison ? kInf - kInf : g[l] / s;
                const double xn = X[i * n + l] + kk[i * n + l] * nu0[i];
                num += (xn - Xc[i * n + l]) * (xn - Xc[i * n + l]);
                den += Xc[i * n + l] * Xc[i * n + l];
                Xc[i * n + l] = xn;
            }
        }
        diff = sqrt(num) / sqrt(den);                            /* :244 */
        ++it;
        if (it == cap) skip = 1;                                 /* :246-248 */
    }
    if (!skip) {
        for (int i = 0; i < m; ++i) {                            /* X = X_corrected; Joseph with the final gain (:250, :297) */
            double AP[MAXN * MAXN], hp[MAXN];
            for (int j = 0; j < n; ++j) {
                X[i * n + j] = Xc[i * n + j];
                hp[j] = 0;
                for (int l = 0; l < n; ++l) hp[j] += h[l] * P[i][l * n + j];
            }
            for (int l = 0; l < n; ++l)
                for (int j = 0; j < n; ++j) AP[l * n + j] = P[i][l * n + j] - kk[i * n + l] * hp[j];
            for (int l = 0; l < n; ++l) {
                double aph = 0;
                for (int j = 0; j < n; ++j) aph += AP[l * n + j] * h[j];
                for (int j = 0; j < n; ++j) P[i][l * n + j] = AP[l * n + j] - aph * kk[i * n + j] + kk[i * n + l] * kk[i * n + j];
            }
        }
    }
    return it;
}

/* One predict + correct of the estimator (experiment.py:166-297) on the row form: f, f_old the noisy features of this and the previous step,
 * h the regressor (the previous command; zeros on the first step), k the step index (annealing).  Updates X and the blocks P in place, fills
 * kappa (the control law's weights, 1 unless GMCKF) and returns the MCKF pass count (0 for the other estimators). */
static int estimator_step(const oracle_params *fp, int k, double *X, double P[][MAXN * MAXN], const double *f, const double *f_old, const double *h,
                          double *kappa) {
    const int m = fp->m, n = fp->n;
    const double sigma = fp->annealing ? fp->kernel_bw + fp->anneal_span * (1.0 - (double)k / fp->k_max) : fp->kernel_bw;
    double nu[MAXM], cs = 1;
    for (int i = 0; i < m; ++i) {
        double pred = 0;
        for (int j = 0; j < n; ++j) pred += X[i * n + j] * h[j];
        nu[i] = (f[i] - f_old[i]) - pred;
    }
    if (fp->method == 4) {
        double ss = 0;
        for (int i = 0; i < m; ++i) ss += nu[i] * nu[i];
        cs = gaussian_kernel(sqrt(ss), sigma);
    }
    if (fp->method == 3) {                                       /* MCKF */
        double Z[MAXM];
        for (int i = 0; i < m; ++i) {
            Z[i] = f[i] - f_old[i];
            kappa[i] = 1;
            for (int j = 0; j < n; ++j) P[i][j * n + j] += 1;
        }
        return mckf_step(m, n, X, P, Z, h, sigma, fp->fpi_threshold, fp->fpi_epoch_max);
    }
    for (int i = 0; i < m; ++i) {
        double g[MAXN], kk[MAXN], a = 0;
        for (int j = 0; j < n; ++j) P[i][j * n + j] += 1;
        for (int l = 0; l < n; ++l) {
            g[l] = 0;
            for (int j = 0; j < n; ++j) g[l] += P[i][l * n + j] * h[j];
        }
        for (int l = 0; l < n; ++l) a += h[l] * g[l];
        double scale = 1, r = 1;
        kappa[i] = 1;
        if (fp->method == 5) { kappa[i] = gaussian_kernel(nu[i], sigma); r = 1.0 / (kappa[i] + fp->reg); }
        if (fp->method == 4) scale = cs;
        const double s = scale * a + r;
        for (int l = 0; l < n; ++l) { kk[l] = scale * g[l] / s; X[i * n + l] += kk[l] * nu[i]; }
        /* Joseph: (I - k h^T) P (I - k h^T)^T + k k^T */
        double AP[MAXN * MAXN], hp[MAXN];
        for (int j = 0; j < n; ++j) {
            hp[j] = 0;
            for (int l = 0; l < n; ++l) hp[j] += h[l] * P[i][l * n + j];
        }
        for (int l = 0; l < n; ++l)
            for (int j = 0; j < n; ++j) AP[l * n + j] = P[i][l * n + j] - kk[l] * hp[j];
        for (int l = 0; l < n; ++l) {
            double aph = 0;
            for (int j = 0; j < n; ++j) aph += AP[l * n + j] * h[j];
            for (int j = 0; j < n; ++j) P[i][l * n + j] = AP[l * n + j] - aph * kk[j] + kk[l] * kk[j];
        }
    }
    return 0;
}

/* One trial.  noise: [K][m] or NULL; x0: [m*n] when !initial_guess.  Outputs (any may be NULL): err [K][m], q_log [K][n],
 * x_log [K][m*n], stats [3], fpi_log [K] (MCKF passes per step).  Returns status (0 success, 1 fail); *k_done receives the number of logged rows. */
int uvs_oracle_closed_loop(const oracle_params *fp, const oracle_plant *pl, const double *q_start, const double *noise, const double *x0,
                           double *err_log, double *q_log, double *x_log, double *stats, int32_t *k_done, int32_t *fpi_log) {
    const int m = fp->m, n = fp->n, K = fp->steps;
    double X[MAXM * MAXN], P[MAXM][MAXN * MAXN], q[MAXN], dq[MAXN], f[MAXM], f_old[MAXM], T[MAXN][16];
    double ise[MAXM] = {0}, iae[MAXM] = {0}, itae[MAXM] = {0};
    for (int j = 0; j < n; ++j) { q[j] = q_start[j]; dq[j] = 0; }
    for (int i = 0; i < m; ++i) {
        for (int e = 0; e < n * n; ++e) P[i][e] = 0;
        for (int j = 0; j < n; ++j) P[i][j * n + j] = 1;
    }
    if (fp->initial_guess) {
        initial_guess(pl, q, m, n, X, f);
    } else {
        for (int i = 0; i < m * n; ++i) X[i] = x0[i];
        for (int i = 0; i < m; ++i) f[i] = 0;
    }
    double t = fp->dt;
    int status = 0, k = 0;
    for (; k < K; ++k) {
        memcpy(f_old, f, sizeof(double) * m);
        if (pl->lin_J) {
            for (int i = 0; i < m; ++i) {
                f[i] = pl->lin_f0[i];
                for (int j = 0; j < n; ++j) f[i] += pl->lin_J[i * n + j] * (q[j] - pl->lin_q0[j]);
            }
        } else {
            fkine_all(pl, q, T);
            features(pl, T[n - 1], f);
        }
        if (noise)
            for (int i = 0; i < m; ++i) f[i] += noise[k * m + i];
        double kappa[MAXM], err[MAXM];
        for (int i = 0; i < m; ++i) err[i] = f[i] - fp->desired[i];
        const int it = estimator_step(fp, k, X, P, f, f_old, dq, kappa);
        if (fpi_log && fp->method == 3) fpi_log[k] = it;
        double y[MAXM], sol[MAXN];
        for (int i = 0; i < m; ++i) y[i] = kappa[i] * err[i];
        if (pinv_apply(X, m, n, y, sol) != 0) { status = 1; break; }
        for (int j = 0; j < n; ++j) dq[j] = -fp->gain * sol[j];
        for (int i = 0; i < m; ++i) {
            if (err_log) err_log[k * m + i] = err[i];
            ise[i] += err[i] * err[i]; iae[i] += fabs(err[i]); itae[i] += t * fabs(err[i]);
        }
        if (q_log) memcpy(q_log + k * n, q, sizeof(double) * n);
        if (x_log) memcpy(x_log + (size_t)k * m * n, X, sizeof(double) * m * n);
        for (int j = 0; j < n; ++j) q[j] += dq[j] * fp->dt;
        t += fp->dt;
    }
    if (stats) {
        double s[3] = {0, 0, 0};
        for (int i = 0; i < m; ++i) { s[0] += ise[i] * ise[i]; s[1] += iae[i] * iae[i]; s[2] += itae[i] * itae[i]; }
        for (int c = 0; c < 3; ++c) stats[c] = sqrt(s[c]);
    }
    *k_done = k;
    return status;
}

/* Open-loop replay of recorded streams (what the reference's loop computes when f and the regressor are given, experiment.py:166-312):
 * f_seq [K+1][m] (row 0 = f_old of the first step), dq_seq [K][n] (row 0 ignored: H = 0 on the first iteration, experiment.py:183), x0 [m*n].
 * Outputs: x_log [K][m*n], cmd_log [K][n] (the command the control law would issue), kappa_log [K][m], fpi_log [K]; any may be NULL.
 * Returns status; *k_done = steps completed before X turned non-finite. */
int uvs_oracle_replay(const oracle_params *fp, const double *f_seq, const double *dq_seq, const double *x0, double *x_log, double *cmd_log,
                      double *kappa_log, int32_t *k_done, int32_t *fpi_log) {
    const int m = fp->m, n = fp->n, K = fp->steps;
    double X[MAXM * MAXN], P[MAXM][MAXN * MAXN], zero[MAXN] = {0};
    for (int i = 0; i < m * n; ++i) X[i] = x0[i];
    for (int i = 0; i < m; ++i) {
        for (int e = 0; e < n * n; ++e) P[i][e] = 0;
        for (int j = 0; j < n; ++j) P[i][j * n + j] = 1;
    }
    int status = 0, k = 0;
    for (; k < K; ++k) {
        const double *f = f_seq + (size_t)(k + 1) * m, *f_old = f_seq + (size_t)k * m;
        double kappa[MAXM], y[MAXM], sol[MAXN];
        const int it = estimator_step(fp, k, X, P, f, f_old, k == 0 ? zero : dq_seq + (size_t)k * n, kappa);
        if (fpi_log && fp->method == 3) fpi_log[k] = it;
        for (int i = 0; i < m; ++i) y[i] = kappa[i] * (f[i] - fp->desired[i]);
        if (pinv_apply(X, m, n, y, sol) != 0) { status = 1; break; }
        if (x_log) memcpy(x_log + (size_t)k * m * n, X, sizeof(double) * m * n);
        if (kappa_log) memcpy(kappa_log + (size_t)k * m, kappa, sizeof(double) * m);
        if (cmd_log)
            for (int j = 0; j < n; ++j) cmd_log[k * n + j] = -fp->gain * sol[j];
    }
    *k_done = k;
    return status;
}

void uvs_oracle_replay_batch(const oracle_params *fp, int64_t T, const double *f_seq, const double *dq_seq, const double *x0, double *x_log,
                             double *cmd_log, double *kappa_log, int32_t *status, int32_t *k_done, int32_t *fpi_log) {
    const size_t K = fp->steps, m = fp->m, n = fp->n;
    for (int64_t t = 0; t < T; ++t)
        status[t] = uvs_oracle_replay(fp, f_seq + t * (K + 1) * m, dq_seq + t * K * n, x0 + t * m * n, x_log ? x_log + t * K * m * n : 0,
                                      cmd_log ? cmd_log + t * K * n : 0, kappa_log ? kappa_log + t * K * m : 0, k_done + t, fpi_log ? fpi_log + t * K : 0);
}

/* Batch driver: trials t = 0..T-1 with per-trial q_start [T][n], noise [T][K][m]; outputs [T][K][...] (may be NULL). */
void uvs_oracle_closed_loop_batch(const oracle_params *fp, const oracle_plant *pl, int64_t T, const double *q_start, const double *noise,
                                  double *err_log, double *q_log, double *x_log, double *stats, int32_t *status, int32_t *k_done, int32_t *fpi_log) {
    const size_t K = fp->steps, m = fp->m, n = fp->n;
    for (int64_t t = 0; t < T; ++t)
        status[t] = uvs_oracle_closed_loop(fp, pl, q_start + t * n, noise ? noise + t * K * m : 0, 0, err_log ? err_log + t * K * m : 0,
                                           q_log ? q_log + t * K * n : 0, x_log ? x_log + t * K * m * n : 0, stats ? stats + 3 * t : 0, k_done + t,
                                           fpi_log ? fpi_log + t * K : 0);
}

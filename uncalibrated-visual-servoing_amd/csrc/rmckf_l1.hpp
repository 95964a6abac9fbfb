// Tuned closed-loop kernel, one filter per lane (lanes_per_filter = 1): the headline path.
//
// Why this shape.  65 536 trials x (8 blocks x 21 doubles of P) is 88 MB -- 69 % of the chip's whole VGPR+AGPR file
// (1024 SIMDs x 512 regs x 64 lanes x 4 B) and more than twice its LDS -- so P must live in registers and there is exactly
// one wavefront per SIMD with nothing to hide latency behind.  Only 256 of the 512 registers are addressable by VALU
// instructions (the AGPR half is reachable through v_accvgpr moves), so everything that is touched once per step moves out:
//   LDS   : X (m*n doubles/lane, [component][lane] so that every ds access is conflict free), the 3*m ISE/IAE/ITAE
//           accumulators                                                        -> 72 x 512 B = 36 KB per wavefront, 4 per CU
//   VGPR  : the block being updated, the regressor, the Householder panel of the control law
//   AGPR  : the other covariance blocks (the compiler parks them there; 84 moves per block per step)
//   SGPR  : per-wavefront stream base pointers; every global access is  s[base] + v_lane_offset  (no per-lane 64-bit math)
// Streams are indexed through uvs_view strides; the trial-fastest layout ([step][component][trial]) makes each of the
// 62 stores and 8 loads per step one contiguous 512-byte wavefront transaction.
#pragma once
#include "rmckf_device.hpp"
#include "rmckf_math.hpp"

namespace uvs {

// Uniform (per-wavefront) base + 32-bit per-lane element offset.
struct LaneStream {
    double *base;          // view base + first_trial_of_wave * trial_stride   (uniform)
    long long sk, sc;      // step / component strides                          (uniform)
    unsigned lo;           // lane * trial_stride                               (per lane)
    bool on;
    UVS_DEV double *row(int k) const { return base + (long long)k * sk; }
};
UVS_DEV LaneStream lane_stream(const View &v, long long wave_first, unsigned lane_in_wave) {
    LaneStream s;
    s.base = v.p + wave_first * v.st;
    s.sk = v.sk;
    s.sc = v.sc;
    s.lo = lane_in_wave * (unsigned)v.st;
    s.on = v.p != nullptr;
    return s;
}

// Plant with the fast bounded-argument sincos (same arithmetic as forward_kinematics<N,false> otherwise).
template <int N>
UVS_DEV void camera_pose_fast(const uvs_plant &pl, const double (&q)[N], double (&rot)[9], double (&pos)[3]) {
    double T[3][4];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s, c;
        sincos_any(q[i] + pl.theta_offset[i], s, c);
        const double ca = pl.cos_alpha[i], sa = pl.sin_alpha[i], aa = pl.a[i], dd = pl.d[i];
        const double l01 = -s * ca, l02 = s * sa, l03 = aa * c;
        const double l11 = c * ca, l12 = -c * sa, l13 = aa * s;
        if (i == 0) {                               // T_0_1 is the first link itself
            T[0][0] = c; T[0][1] = l01; T[0][2] = l02; T[0][3] = l03;
            T[1][0] = s; T[1][1] = l11; T[1][2] = l12; T[1][3] = l13;
            T[2][0] = 0.0; T[2][1] = sa; T[2][2] = ca; T[2][3] = dd;
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double t0 = T[r][0], t1 = T[r][1], t2 = T[r][2], t3 = T[r][3];
                T[r][0] = fma(t0, c, t1 * s);
                T[r][1] = fma(t0, l01, fma(t1, l11, t2 * sa));
                T[r][2] = fma(t0, l02, fma(t1, l12, t2 * ca));
                T[r][3] = fma(t0, l03, fma(t1, l13, fma(t2, dd, t3)));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) rot[3 * r + cidx] = T[r][cidx];
        pos[r] = T[r][3];
    }
}

// Householder QR least squares on an M x (N+1) panel held by one lane (M >= N); fast reciprocal / rsqrt.
template <int M, int N>
UVS_DEV void lstsq_tall_l1(double (&a)[M][N + 1], double (&sol)[N]) {
    double rdiag[N];                                   // 1 / R_cc (0 marks a zero column)
#pragma unroll
    for (int c = 0; c < N; ++c) {
        double sig = 0.0;
#pragma unroll
        for (int r = c + 1; r < M; ++r) sig = fma(a[r][c], a[r][c], sig);
        const double piv = a[c][c];
        const double n2 = fma(piv, piv, sig);
        double nrm, rn;
        fast_sqrt_rsqrt(n2, nrm, rn);
        const bool zero = !(n2 > 0.0);                 // all-zero column (also catches NaN: result stays NaN downstream)
        const double alpha = (piv >= 0.0) ? -nrm : nrm;
        const double vp = piv - alpha;
        // tau = 1 / (nrm (nrm + |piv|)) = 1 / (-alpha vp)
        const double tau = zero ? 0.0 : fast_rcp(-alpha * vp);
#pragma unroll
        for (int j = c + 1; j <= N; ++j) {
            double d = vp * a[c][j];
#pragma unroll
            for (int r = c + 1; r < M; ++r) d = fma(a[r][c], a[r][j], d);
            d *= tau;
            a[c][j] = fma(-d, vp, a[c][j]);
#pragma unroll
            for (int r = c + 1; r < M; ++r) a[r][j] = fma(-d, a[r][c], a[r][j]);
        }
        rdiag[c] = zero ? 0.0 : fast_rcp(alpha);
    }
#pragma unroll
    for (int c = N - 1; c >= 0; --c) {
        double rhs = a[c][N];
#pragma unroll
        for (int j = c + 1; j < N; ++j) rhs = fma(-a[c][j], sol[j], rhs);
        sol[c] = rhs * rdiag[c];
    }
}

template <int M, int N, int METHOD, int PLANT>
__global__ __launch_bounds__(64) void closed_loop_l1_kernel(const ClosedArgs A) {
    static_assert(M >= N, "the tuned kernel covers tall Jacobians; wide ones use the generic path");
    constexpr int NP = Sym<N>::NP;
    __shared__ double lds_x[M * N][64];
    __shared__ double lds_acc[3 * M][64];

    const unsigned lane = threadIdx.x;
    const long long wave_first = (long long)blockIdx.x * 64;
    const bool valid = wave_first + lane < A.T;
    const unsigned lt = valid ? lane : (unsigned)(A.T - 1 - wave_first);        // padding lanes shadow the last trial
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;

    const LaneStream s_noise = lane_stream(A.noise, wave_first, lt), s_x = lane_stream(A.x_out, wave_first, lt),
                     s_err = lane_stream(A.err_out, wave_first, lt), s_q = lane_stream(A.q_out, wave_first, lt),
                     s_f = lane_stream(A.f_out, wave_first, lt), s_dq = lane_stream(A.dq_out, wave_first, lt);

    double q[N], dq[N], f_prev[M];
    double p[M][NP];
    {
        const LaneStream s_q0 = lane_stream(A.q_start, wave_first, lt);
#pragma unroll
        for (int j = 0; j < N; ++j) { q[j] = (s_q0.base + j * s_q0.sc)[s_q0.lo]; dq[j] = 0.0; }
        double x0[M][N];
        if (fp.initial_guess) {
            initial_guess<M, N, 1>(A.plant, q, 0, x0, f_prev);
        } else {
            const LaneStream s_x0 = lane_stream(A.x0, wave_first, lt);
#pragma unroll
            for (int i = 0; i < M; ++i) {
                f_prev[i] = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) x0[i][j] = (s_x0.base + (i * N + j) * s_x0.sc)[s_x0.lo];
            }
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) lds_x[i * N + j][lane] = x0[i][j];
#pragma unroll
        for (int i = 0; i < 3 * M; ++i) lds_acc[i][lane] = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = l; j < N; ++j) p[i][Sym<N>::at(l, j)] = (l == j) ? 1.0 : 0.0;
    }

    double t = fp.dt;
    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true;

    for (int k = 0; k < K; ++k) {
        // ---- measurement: noise load first (latency hides under the kinematics), plant, innovation inputs
        double nz[M];
        {
            double *nb = s_noise.row(k);
#pragma unroll
            for (int i = 0; i < M; ++i) nz[i] = s_noise.on ? (nb + i * s_noise.sc)[s_noise.lo] : 0.0;
        }
        double z[M];
        if constexpr (PLANT == UVS_PLANT_LINEAR) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double acc = A.plant.lin_f0[i];
#pragma unroll
                for (int j = 0; j < N; ++j) acc = fma(A.plant.lin_jacobian[i * N + j], q[j] - A.plant.lin_q0[j], acc);
                z[i] = acc;
            }
        } else {
            double rot[9], pos[3];
            camera_pose_fast<N>(A.plant, q, rot, pos);
#pragma unroll
            for (int pt = 0; pt < M / 2; ++pt) {
                const double *w = A.plant.points[pt];
                const double dx = w[0] - pos[0], dy = w[1] - pos[1], dz = w[2] - pos[2];
                const double xc = fma(rot[0], dx, fma(rot[3], dy, rot[6] * dz));
                const double yc = fma(rot[1], dx, fma(rot[4], dy, rot[7] * dz));
                const double zc = fma(rot[2], dx, fma(rot[5], dy, rot[8] * dz));
                const double iz = fast_rcp(zc);
                z[2 * pt] = fma(A.plant.focal * xc, iz, A.plant.center);
                z[2 * pt + 1] = fma(A.plant.focal * yc, iz, A.plant.center);
            }
        }
        const double sigma = bandwidth(fp, k);
        const double neg_half_inv_s2 = -0.5 * fast_rcp(sigma * sigma);
        double *xb = s_x.row(k);
        double kap[M];
        double chk = 0.0;                                        // turns NaN as soon as any state entry is non-finite
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const double fi = z[i] + nz[i];                      // noisy feature (experiment.py:134-135)
            const double zi = fi - f_prev[i];                    // measurement Z (experiment.py:170-177)
            f_prev[i] = fi;
            double x[N], g[N];
#pragma unroll
            for (int j = 0; j < N; ++j) x[j] = lds_x[i * N + j][lane];
            double pred = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) pred = fma(x[j], dq[j], pred);
            const double nu = zi - pred;
#pragma unroll
            for (int l = 0; l < N; ++l) p[i][Sym<N>::at(l, l)] += 1.0;
#pragma unroll
            for (int l = 0; l < N; ++l) {
                double acc = p[i][Sym<N>::at(l, 0)] * dq[0];
#pragma unroll
                for (int j = 1; j < N; ++j) acc = fma(p[i][Sym<N>::at(l, j)], dq[j], acc);
                g[l] = acc;
            }
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) a = fma(dq[l], g[l], a);
            double gamma;
            if constexpr (METHOD == UVS_METHOD_GMCKF) {
                kap[i] = exp((nu * nu) * neg_half_inv_s2);
                const double d = kap[i] + fp.reg;                // gamma = 1 / (a + 1/d) = d / (a d + 1)
                gamma = d * fast_rcp(fma(a, d, 1.0));
            } else {                                             // KF (IMCCKF runs on the generic path)
                kap[i] = 1.0;
                gamma = fast_rcp(a + 1.0);
            }
            const double step = gamma * nu;
            const double beta = gamma * (2.0 - gamma * (a + 1.0));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                x[j] = fma(g[j], step, x[j]);
                chk = fma(x[j], 0.0, chk);
                lds_x[i * N + j][lane] = x[j];
            }
            if (s_x.on && alive && valid) {
#pragma unroll
                for (int j = 0; j < N; ++j) (xb + (i * N + j) * s_x.sc)[s_x.lo] = x[j];
            }
#pragma unroll
            for (int l = 0; l < N; ++l) {
                const double w = beta * g[l];
#pragma unroll
                for (int j = l; j < N; ++j) p[i][Sym<N>::at(l, j)] = fma(-w, g[j], p[i][Sym<N>::at(l, j)]);
            }
        }
        // NB: the X rows of step k were stored before the FAIL test of step k; a failing trial reports k_done = k and
        // callers ignore rows >= k_done (the reference breaks before logging row k, experiment.py:313-316).
        if (alive && !(chk == 0.0)) {
            alive = false;
            status = UVS_STATUS_FAIL;
            k_done = k;
        }
        if (!__any(alive)) break;

        // ---- control law: dq = -gain * pinv(X) (kappa o err)
        {
            double panel[M][N + 1];
#pragma unroll
            for (int i = 0; i < M; ++i) {
#pragma unroll
                for (int j = 0; j < N; ++j) panel[i][j] = lds_x[i * N + j][lane];
                panel[i][N] = kap[i] * (f_prev[i] - fp.desired[i]);
            }
            double sol[N];
            lstsq_tall_l1<M, N>(panel, sol);
#pragma unroll
            for (int j = 0; j < N; ++j) dq[j] = -fp.gain * sol[j];
        }

        // ---- logs and statistics
        if (alive && valid) {
            double *eb = s_err.row(k), *fb = s_f.row(k), *qb = s_q.row(k), *db = s_dq.row(k);
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const double e = f_prev[i] - fp.desired[i];
                if (s_err.on) (eb + i * s_err.sc)[s_err.lo] = e;
                if (s_f.on) (fb + i * s_f.sc)[s_f.lo] = f_prev[i];
                const double ae = fabs(e);
                lds_acc[i][lane] = fma(e, e, lds_acc[i][lane]);
                lds_acc[M + i][lane] += ae;
                lds_acc[2 * M + i][lane] = fma(t, ae, lds_acc[2 * M + i][lane]);
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (s_q.on) (qb + j * s_q.sc)[s_q.lo] = q[j];
                if (s_dq.on) (db + j * s_dq.sc)[s_dq.lo] = dq[j];
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) q[j] = fma(dq[j], fp.dt, q[j]);
        t += fp.dt;
    }

    if (!valid) return;
    const long long trial = wave_first + lane;
    double s2[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = lds_acc[c * M + i][lane];
            s2[c] = fma(v, v, s2[c]);
        }
    }
    if (A.stats) {
#pragma unroll
        for (int c = 0; c < 3; ++c) A.stats[3 * trial + c] = sqrt(s2[c]);
    }
    if (A.status) A.status[trial] = status;
    if (A.k_done) A.k_done[trial] = k_done;
    if (A.x_final.on()) {
#pragma unroll
        for (int c = 0; c < M * N; ++c) *A.x_final.at(trial, 0, c) = lds_x[c][lane];
    }
    if (A.p_final.on()) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = 0; j < N; ++j) *A.p_final.at(trial, 0, (i * N + l) * N + j) = p[i][Sym<N>::at(l, j)];
    }
}

}  // namespace uvs

// Tuned closed-loop kernel: L = 1 or 2 lanes per filter, the headline path (BASELINE config 2).
//
// Sizing.  65 536 trials x (8 blocks x 21 doubles of P) is 88 MB -- 69 % of the chip's whole VGPR+AGPR file
// (1024 SIMDs x 512 regs x 64 lanes x 4 B) and twice its LDS -- so P lives in registers, at most one wavefront fits per
// SIMD and nothing hides latency: any scratch spill is a full memory round trip on the critical path (measured: 250 scratch
// loads per step cost 7x the arithmetic).  Only 256 of the 512 registers are VALU-addressable.  Hence:
//   * L lanes share a filter, lane s owning rows s, s+L, s+2L, ... (interleaved, so both lanes of a pair run the same
//     static Householder code); per lane P is (m/L) x 21 doubles in VGPRs;
//   * X ((m/L)*n doubles per lane) and the ISE/IAE/ITAE accumulators live in LDS as [component][lane] (conflict free);
//   * the pair exchanges partial sums / pivots with DPP quad_perm moves (no LDS, no ds_bpermute);
//   * every global access is  s[wavefront base] + v[lane offset]: stream bases advance in SGPRs;
//   * divisions / roots / sincos come from rmckf_math.hpp.
// With the trial-fastest layout ([step][component][trial]) each wavefront store covers 512 / L contiguous bytes per row set.
#pragma once
#include "rmckf_device.hpp"
#include "rmckf_math.hpp"

namespace uvs {

// DPP quad_perm move of a double (two 32-bit moves).  CTRL = a | b<<2 | c<<4 | d<<6 selects the source lane of each lane of a quad.
template <int CTRL>
UVS_DEV double dpp_quad(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int kSwapPair = 0xB1;      // quad_perm [1,0,3,2]: partner lane
constexpr int kFromEven = 0xA0;      // quad_perm [0,0,2,2]: value of the pair's even lane
constexpr int kFromOdd = 0xF5;       // quad_perm [1,1,3,3]: value of the pair's odd lane

template <int L>
UVS_DEV double pair_sum(double v) {
    if constexpr (L == 1) return v;
    return v + dpp_quad<kSwapPair>(v);
}
template <int L, int OWNER>
UVS_DEV double pair_from(double v) {
    if constexpr (L == 1) return v;
    return OWNER ? dpp_quad<kFromOdd>(v) : dpp_quad<kFromEven>(v);
}

// Uniform (per-wavefront) base + 32-bit per-lane element offset.
struct LaneStream {
    double *base;          // view base + first_trial_of_wave * trial_stride   (uniform)
    long long sk, sc;      // step / component strides                          (uniform)
    bool on;
    UVS_DEV double *row(int k) const { return base + (long long)k * sk; }
};
UVS_DEV LaneStream lane_stream(const View &v, long long wave_first_trial) {
    return LaneStream{v.p + wave_first_trial * v.st, v.sk, v.sc, v.p != nullptr};
}

// Camera pose with the fast bounded-argument sincos (same arithmetic as forward_kinematics<N,false> otherwise).
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

// Householder QR least squares, rows interleaved over the L lanes of a filter: local row r of lane s is global row r*L + s.
// In column c the local row m = c / L is the pivot row on lane c % L, an ordinary "below" row on lanes > c % L and already
// finished on lanes < c % L; rows r > m are below the pivot on every lane -- so all lanes run the same unrolled code and only
// the treatment of row m is selected per lane.
template <int M, int N, int L>
UVS_DEV void lstsq_tall_tuned(double (&a)[M / L][N + 1], int sub, double (&sol)[N]) {
    constexpr int R = M / L;
    double rdiag[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        constexpr int dummy = 0; (void)dummy;
        const int m = c / L, owner = c % L;
        const bool is_piv = (L == 1) || (sub == owner);
        const bool is_below = (L > 1) && (sub > owner);
        double sig = is_below ? a[m][c] * a[m][c] : 0.0;
#pragma unroll
        for (int r = m + 1; r < R; ++r) sig = fma(a[r][c], a[r][c], sig);
        sig = pair_sum<L>(sig);
        const double piv = (owner == 0) ? pair_from<L, 0>(a[m][c]) : pair_from<L, 1>(a[m][c]);
        const double n2 = fma(piv, piv, sig);
        double nrm, rn;
        fast_sqrt_rsqrt(n2, nrm, rn);
        const bool zero = !(n2 > 0.0);
        const double alpha = (piv >= 0.0) ? -nrm : nrm;
        const double vp = piv - alpha;
        const double tau = zero ? 0.0 : fast_rcp(-alpha * vp);          // 2 / (v.v)
        const double vm = is_piv ? vp : (is_below ? a[m][c] : 0.0);     // this lane's entry of the Householder vector in row m
#pragma unroll
        for (int j = c + 1; j <= N; ++j) {
            double d = vm * a[m][j];
#pragma unroll
            for (int r = m + 1; r < R; ++r) d = fma(a[r][c], a[r][j], d);
            d = pair_sum<L>(d) * tau;
            a[m][j] = fma(-d, vm, a[m][j]);
#pragma unroll
            for (int r = m + 1; r < R; ++r) a[r][j] = fma(-d, a[r][c], a[r][j]);
        }
        rdiag[c] = zero ? 0.0 : fast_rcp(alpha);
    }
#pragma unroll
    for (int c = N - 1; c >= 0; --c) {
        const int m = c / L, owner = c % L;
        double rhs = a[m][N];
#pragma unroll
        for (int j = c + 1; j < N; ++j) rhs = fma(-a[m][j], sol[j], rhs);
        rhs = (owner == 0) ? pair_from<L, 0>(rhs) : pair_from<L, 1>(rhs);
        sol[c] = rhs * rdiag[c];
    }
}

template <int M, int N, int L, int METHOD, int PLANT>
__global__ __launch_bounds__(64) void closed_loop_tuned_kernel(const ClosedArgs A) {
    static_assert(M >= N && (L == 1 || L == 2) && M % L == 0, "tuned kernel: tall Jacobian, 1 or 2 lanes per filter");
    constexpr int R = M / L, NP = Sym<N>::NP, TPW = 64 / L;       // rows per lane, packed block size, trials per wavefront
    __shared__ double lds_x[R * N][64];
    __shared__ double lds_acc[3 * R][64];

    const unsigned lane = threadIdx.x;
    const int sub = (L == 1) ? 0 : (int)(lane & (L - 1));
    const long long wave_first = (long long)blockIdx.x * TPW;      // first trial of this wavefront (uniform)
    const unsigned tl = lane / L;                                   // trial within the wavefront
    const bool valid = wave_first + tl < A.T;
    const unsigned tq = valid ? tl : (unsigned)(A.T - 1 - wave_first);   // padding lanes shadow the last trial
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;

    const LaneStream s_noise = lane_stream(A.noise, wave_first), s_x = lane_stream(A.x_out, wave_first),
                     s_err = lane_stream(A.err_out, wave_first), s_q = lane_stream(A.q_out, wave_first),
                     s_f = lane_stream(A.f_out, wave_first), s_dq = lane_stream(A.dq_out, wave_first);
    // per-lane element offsets: trial * trial_stride + (first owned row) * component stride
    const unsigned lo_noise = tq * (unsigned)A.noise.st + sub * (unsigned)A.noise.sc;
    const unsigned lo_err = tq * (unsigned)A.err_out.st + sub * (unsigned)A.err_out.sc;
    const unsigned lo_f = tq * (unsigned)A.f_out.st + sub * (unsigned)A.f_out.sc;
    const unsigned lo_x = tq * (unsigned)A.x_out.st + sub * N * (unsigned)A.x_out.sc;
    const unsigned lo_q = tq * (unsigned)A.q_out.st, lo_dq = tq * (unsigned)A.dq_out.st;

    double q[N], dq[N], f_prev[R], des[R];
    double p[R][NP];
    {
#pragma unroll
        for (int j = 0; j < N; ++j) { q[j] = *A.q_start.at(wave_first + tq, 0, j); dq[j] = 0.0; }
        double x0[R][N];
        if (fp.initial_guess) {
            // generic helper owns rows sub*R .. ; here rows are interleaved, so evaluate row by row
            double xa[M][N], fa[M];
            initial_guess<M, N, 1>(A.plant, q, 0, xa, fa);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f_prev[r] = (L == 1) ? fa[r] : (sub ? fa[r * L + 1] : fa[r * L]);
#pragma unroll
                for (int j = 0; j < N; ++j) x0[r][j] = (L == 1) ? xa[r][j] : (sub ? xa[r * L + 1][j] : xa[r * L][j]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f_prev[r] = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) x0[r][j] = *A.x0.at(wave_first + tq, 0, (r * L + sub) * N + j);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            des[r] = (L == 1) ? fp.desired[r] : (sub ? fp.desired[r * L + 1] : fp.desired[r * L]);
#pragma unroll
            for (int j = 0; j < N; ++j) lds_x[r * N + j][lane] = x0[r][j];
        }
#pragma unroll
        for (int i = 0; i < 3 * R; ++i) lds_acc[i][lane] = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = l; j < N; ++j) p[r][Sym<N>::at(l, j)] = (l == j) ? 1.0 : 0.0;
    }

    double t = fp.dt;
    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true;

    for (int k = 0; k < K; ++k) {
        // ---- noise load first (its latency hides under the kinematics)
        double nz[R];
        {
            double *nb = s_noise.row(k);
#pragma unroll
            for (int r = 0; r < R; ++r) nz[r] = s_noise.on ? (nb + r * L * s_noise.sc)[lo_noise] : 0.0;
        }
        // ---- plant: noise-free features of this lane's rows
        double z[R];
        if constexpr (PLANT == UVS_PLANT_LINEAR) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = r * L + sub;
                double acc = A.plant.lin_f0[row];
#pragma unroll
                for (int j = 0; j < N; ++j) acc = fma(A.plant.lin_jacobian[row * N + j], q[j] - A.plant.lin_q0[j], acc);
                z[r] = acc;
            }
        } else {
            double rot[9], pos[3];
            camera_pose_fast<N>(A.plant, q, rot, pos);
            if constexpr (L == 1) {
#pragma unroll
                for (int pt = 0; pt < M / 2; ++pt) {
                    const double *w = A.plant.points[pt];
                    const double dx = w[0] - pos[0], dy = w[1] - pos[1], dz = w[2] - pos[2];
                    const double xc = fma(rot[0], dx, fma(rot[3], dy, rot[6] * dz));
                    const double yc = fma(rot[1], dx, fma(rot[4], dy, rot[7] * dz));
                    const double iz = fast_rcp(fma(rot[2], dx, fma(rot[5], dy, rot[8] * dz)));
                    z[2 * pt] = fma(A.plant.focal * xc, iz, A.plant.center);
                    z[2 * pt + 1] = fma(A.plant.focal * yc, iz, A.plant.center);
                }
            } else {
                // lane 0 owns the u rows, lane 1 the v rows of every point: pick the camera axis once
                const double ax = sub ? rot[1] : rot[0], ay = sub ? rot[4] : rot[3], az = sub ? rot[7] : rot[6];
#pragma unroll
                for (int pt = 0; pt < R; ++pt) {
                    const double *w = A.plant.points[pt];
                    const double dx = w[0] - pos[0], dy = w[1] - pos[1], dz = w[2] - pos[2];
                    const double ic = fma(ax, dx, fma(ay, dy, az * dz));
                    const double iz = fast_rcp(fma(rot[2], dx, fma(rot[5], dy, rot[8] * dz)));
                    z[pt] = fma(A.plant.focal * ic, iz, A.plant.center);
                }
            }
        }
        const double sigma = bandwidth(fp, k);
        const double neg_half_inv_s2 = -0.5 * fast_rcp(sigma * sigma);
        double *xb = s_x.row(k);
        double kap[R];
        double chk = 0.0;                                        // turns NaN as soon as any state entry is non-finite
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double fi = z[r] + nz[r];                      // noisy feature (experiment.py:134-135)
            const double zi = fi - f_prev[r];                    // measurement Z (experiment.py:170-177)
            f_prev[r] = fi;
            double x[N], g[N];
#pragma unroll
            for (int j = 0; j < N; ++j) x[j] = lds_x[r * N + j][lane];
            double pred = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) pred = fma(x[j], dq[j], pred);
            const double nu = zi - pred;                         // innovation (experiment.py:274)
#pragma unroll
            for (int l = 0; l < N; ++l) p[r][Sym<N>::at(l, l)] += 1.0;           // P + Q (experiment.py:167)
#pragma unroll
            for (int l = 0; l < N; ++l) {
                double acc = p[r][Sym<N>::at(l, 0)] * dq[0];
#pragma unroll
                for (int j = 1; j < N; ++j) acc = fma(p[r][Sym<N>::at(l, j)], dq[j], acc);
                g[l] = acc;
            }
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) a = fma(dq[l], g[l], a);
            double gamma;
            if constexpr (METHOD == UVS_METHOD_GMCKF) {
                kap[r] = exp((nu * nu) * neg_half_inv_s2);       // utils.py:171-172
                const double d = kap[r] + fp.reg;                // gamma = 1 / (a + 1/d) = d / (a d + 1) (experiment.py:280-286)
                gamma = d * fast_rcp(fma(a, d, 1.0));
            } else {                                             // KF (experiment.py:192)
                kap[r] = 1.0;
                gamma = fast_rcp(a + 1.0);
            }
            const double step = gamma * nu;
            const double beta = gamma * (2.0 - gamma * (a + 1.0));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                x[j] = fma(g[j], step, x[j]);                    // X + K (Z - H X) (experiment.py:291)
                chk = fma(x[j], 0.0, chk);
                lds_x[r * N + j][lane] = x[j];
            }
            if (s_x.on && alive && valid) {
#pragma unroll
                for (int j = 0; j < N; ++j) (xb + (r * L * N + j) * s_x.sc)[lo_x] = x[j];
            }
#pragma unroll
            for (int l = 0; l < N; ++l) {                        // Joseph update with R = 1: P -= beta g g^T
                const double w = beta * g[l];
#pragma unroll
                for (int j = l; j < N; ++j) p[r][Sym<N>::at(l, j)] = fma(-w, g[j], p[r][Sym<N>::at(l, j)]);
            }
        }
        // NB: the X rows of step k are stored before its FAIL test; a failing trial reports k_done = k and callers ignore rows
        // >= k_done (the reference breaks before logging row k, experiment.py:313-316).
        chk = pair_sum<L>(chk);
        if (alive && !(chk == 0.0)) {
            alive = false;
            status = UVS_STATUS_FAIL;
            k_done = k;
        }
        if (!__any(alive)) break;

        // ---- control law: dq = -gain * pinv(X) (kappa o err) (experiment.py:300-312)
        {
            double panel[R][N + 1];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < N; ++j) panel[r][j] = lds_x[r * N + j][lane];
                panel[r][N] = kap[r] * (f_prev[r] - des[r]);
            }
            double sol[N];
            lstsq_tall_tuned<M, N, L>(panel, sub, sol);
#pragma unroll
            for (int j = 0; j < N; ++j) dq[j] = -fp.gain * sol[j];
        }

        // ---- logs and statistics
        if (alive && valid) {
            double *eb = s_err.row(k), *fb = s_f.row(k), *qb = s_q.row(k), *db = s_dq.row(k);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double e = f_prev[r] - des[r];             // experiment.py:302
                if (s_err.on) (eb + r * L * s_err.sc)[lo_err] = e;
                if (s_f.on) (fb + r * L * s_f.sc)[lo_f] = f_prev[r];
                const double ae = fabs(e);
                lds_acc[r][lane] = fma(e, e, lds_acc[r][lane]);
                lds_acc[R + r][lane] += ae;
                lds_acc[2 * R + r][lane] = fma(t, ae, lds_acc[2 * R + r][lane]);
            }
            if (sub == 0) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (s_q.on) (qb + j * s_q.sc)[lo_q] = q[j];
                    if (s_dq.on) (db + j * s_dq.sc)[lo_dq] = dq[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) q[j] = fma(dq[j], fp.dt, q[j]);            // new_q = q + dq t_s (experiment.py:320)
        t += fp.dt;
    }

    double s2[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = lds_acc[c * R + r][lane];
            s2[c] = fma(v, v, s2[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) s2[c] = pair_sum<L>(s2[c]);
    if (!valid) return;
    const long long trial = wave_first + tl;
    if (sub == 0) {
        if (A.stats) {
#pragma unroll
            for (int c = 0; c < 3; ++c) A.stats[3 * trial + c] = sqrt(s2[c]);
        }
        if (A.status) A.status[trial] = status;
        if (A.k_done) A.k_done[trial] = k_done;
    }
    if (A.x_final.on()) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) *A.x_final.at(trial, 0, (r * L + sub) * N + j) = lds_x[r * N + j][lane];
    }
    if (A.p_final.on()) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = 0; j < N; ++j) *A.p_final.at(trial, 0, ((r * L + sub) * N + l) * N + j) = p[r][Sym<N>::at(l, j)];
    }
}

}  // namespace uvs

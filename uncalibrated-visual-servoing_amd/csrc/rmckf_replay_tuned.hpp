// Tuned replay kernel: the estimator + control law of experiment.py:166-312 over recorded feature / joint-delta streams, i.e. exactly the
// per-step traffic north_star prices (read f and dq, write X and err: 8 (2m + n + mn) B per update).  Two lanes per filter with the
// register / LDS residency of the tuned closed-loop kernel (rmckf_tuned.hpp): interleaved rows (row = 2 r + sub), PV covariance blocks
// of a lane in VGPRs and the rest in LDS, X in LDS, Householder least squares across the pair.  What differs from the closed loop:
//   * no plant -- the next step's f and dq are fetched one whole step ahead (a lone wavefront cannot hide HBM latency otherwise);
//   * the control law is compiled out (CMD = false) when the caller does not ask for the commanded dq: the estimator alone;
//   * no statistics.
// Store discipline as in the closed loop: straight-line step body, padding lanes shadow the last trial, a FAILed trial keeps running on
// NaNs and its rows at and after k_done are unspecified (the generic replay_kernel leaves them untouched instead).
#pragma once
#include "rmckf_tuned.hpp"

namespace uvs {

template <int M, int N, int METHOD, int PV, bool XOUT, bool CMD>
__global__ __launch_bounds__(64) void replay_tuned_kernel(const ReplayArgs A) {
    static_assert(M % 2 == 0 && M >= N && N % 2 == 0, "tuned replay: tall Jacobian, 2 lanes per filter");
    constexpr int L = 2, R = M / L, NP = Sym<N>::NP, TPW = 64 / L, JG = N / 2;
    static_assert(PV >= 0 && PV <= R, "PV counts covariance blocks");
    constexpr int PL = R - PV;
    __shared__ double lds_x[R * N][64];
    __shared__ double lds_p[PL > 0 ? PL * NP : 1][64];

    const unsigned lane = threadIdx.x;
    const int sub = (int)(lane & 1);
    const long long wave_first = (long long)blockIdx.x * TPW;
    const unsigned tl = lane / L;
    const bool valid = wave_first + tl < A.T;
    const long long trial = valid ? wave_first + tl : A.T - 1;     // padding lanes shadow the last trial
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;

    // stream cursors: inputs point one step ahead of the step being computed
    const double *pf = A.f.at(trial, 1, sub);                     // f_{k+1} of step k, rows sub, sub + 2, ...
    const double *pd = A.dq.at(trial, 1, 0);                      // regressor of step k + 1
    double *px = (XOUT && A.x_out.p) ? A.x_out.at(trial, 0, sub * N) : nullptr;
    double *pe = A.err_out.p ? A.err_out.at(trial, 0, sub) : nullptr;
    double *pk = A.kappa_out.p ? A.kappa_out.at(trial, 0, sub) : nullptr;
    double *pc = (CMD && A.dqcmd_out.p) ? A.dqcmd_out.at(trial, 0, sub * JG) : nullptr;
    const bool on_err = A.err_out.p != nullptr, on_kappa = A.kappa_out.p != nullptr;

    double f_prev[R], des[R], f_next[R], h_next[N];
    double p[PV > 0 ? PV : 1][NP];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        des[r] = pick_sub<L>(&fp.desired[r * L], sub);
        f_prev[r] = *A.f.at(trial, 0, r * L + sub);
        f_next[r] = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) lds_x[r * N + j][lane] = *A.x0.at(trial, 0, (r * L + sub) * N + j);
#pragma unroll
        for (int l = 0; l < N; ++l)
#pragma unroll
            for (int j = l; j < N; ++j) {
                const double v = (l == j) ? 1.0 : 0.0;             // P = I (experiment.py:73)
                if (r < PV) p[r < PV ? r : 0][Sym<N>::at(l, j)] = v;
                else lds_p[(r - PV) * NP + Sym<N>::at(l, j)][lane] = v;
            }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) h_next[j] = 0.0;                    // first_run: H = 0 (experiment.py:183-185)
    if (K > 0) {
        const double *pr = pf;
#pragma unroll
        for (int r = 0; r < R; ++r) { f_next[r] = *pr; pr += L * A.f.sc; }
        pf += A.f.sk;
    }

    // Drain the loads above before the loop: otherwise the loop header inherits "f_next may still be in flight" from the entry edge
    // and the compiler's conservative s_waitcnt there (vmcnt counts in order) also waits for the previous step's stores, every step.
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0)

    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true, flagged = false;                          // flagged: rank-deficient Jacobian seen -> careful second pass
    for (int k = 0; k < K; ++k) {
        double f[R], dq[N];
#pragma unroll
        for (int r = 0; r < R; ++r) f[r] = f_next[r];
#pragma unroll
        for (int j = 0; j < N; ++j) dq[j] = h_next[j];
        if (k + 1 < K) {                                         // inputs of step k + 1: a whole step to arrive
            const double *pr = pf;
#pragma unroll
            for (int r = 0; r < R; ++r) { f_next[r] = *pr; pr += L * A.f.sc; }
            const double *pj = pd;
#pragma unroll
            for (int j = 0; j < N; ++j) { h_next[j] = *pj; pj += A.dq.sc; }
            pf += A.f.sk;
            pd += A.dq.sk;
        }
        const double sigma = bandwidth(fp, k);
        const double neg_half_inv_s2 = -0.5 * fast_rcp(sigma * sigma);
        double c_shared = 1.0;
        if constexpr (METHOD == UVS_METHOD_IMCCKF) {             // one weight for the whole filter (experiment.py:258-261)
            double ss = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double pred = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) pred = fma(lds_x[r * N + j][lane], dq[j], pred);
                const double nu = (f[r] - f_prev[r]) - pred;
                ss = fma(nu, nu, ss);
            }
            c_shared = exp_nonpos(pair_sum<L>(ss) * neg_half_inv_s2);
        }
        double kap[R], err[R];
        double chk = 0.0;                                        // turns NaN as soon as any state entry is non-finite
        FpiProbe fpi;
        if constexpr (METHOD == UVS_METHOD_MCKF) {
            mckf_underflow_prepass<R>(fpi, [&](int r) {
                double pred = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) pred = fma(lds_x[r * N + j][lane], dq[j], pred);
                const double nu = (f[r] - f_prev[r]) - pred;
                return (nu * nu) * neg_half_inv_s2;
            });
            fpi.skip = pair_sum<L>(fpi.skip ? 1.0 : 0.0) != 0.0;
            fpi.unsure = pair_sum<L>(fpi.unsure ? 1.0 : 0.0) != 0.0;
        }
        double *pxr = px;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double zi = f[r] - f_prev[r];                  // measurement Z (experiment.py:170-177)
            f_prev[r] = f[r];
            err[r] = f[r] - des[r];                              // experiment.py:302
            double x[N], pb[NP];
#pragma unroll
            for (int j = 0; j < N; ++j) x[j] = lds_x[r * N + j][lane];
#pragma unroll
            for (int e = 0; e < NP; ++e) pb[e] = (r < PV) ? p[r < PV ? r : 0][e] : lds_p[(r >= PV ? r - PV : 0) * NP + e][lane];
            rmckf_row<N, METHOD>(x, pb, dq, zi, neg_half_inv_s2, c_shared, fp.reg, kap[r], chk, fpi);
#pragma unroll
            for (int j = 0; j < N; ++j) lds_x[r * N + j][lane] = x[j];
            if constexpr (XOUT) {
                double *pcx = pxr;
#pragma unroll
                for (int j = 0; j < N; ++j) { *pcx = x[j]; pcx += A.x_out.sc; }
                pxr += L * N * A.x_out.sc;
            }
#pragma unroll
            for (int e = 0; e < NP; ++e) {
                if (r < PV) p[r < PV ? r : 0][e] = pb[e];
                else lds_p[(r >= PV ? r - PV : 0) * NP + e][lane] = pb[e];
            }
        }
        if constexpr (XOUT) px += A.x_out.sk;
        asm volatile("" ::: "memory");                           // LDS is the only copy of X from here on
        chk = pair_sum<L>(chk);
        if (alive && !(chk == 0.0)) {                            // pinv would raise (experiment.py:313-316)
            alive = false;
            status = UVS_STATUS_FAIL;
            k_done = k;
        }
        if (!__any(alive)) break;
        if constexpr (METHOD == UVS_METHOD_MCKF) {               // first fixed-point pass not conclusive: the careful second pass redoes the trial
            fpi.num = pair_sum<L>(fpi.num);
            fpi.den = pair_sum<L>(fpi.den);
            flagged |= alive && fpi_needs_more(fpi, fp);
        }

        if (on_err) {
            double *po = pe;
#pragma unroll
            for (int r = 0; r < R; ++r) { *po = err[r]; po += L * A.err_out.sc; }
            pe += A.err_out.sk;
        }
        if (on_kappa) {
            double *po = pk;
#pragma unroll
            for (int r = 0; r < R; ++r) { *po = kap[r]; po += L * A.kappa_out.sc; }
            pk += A.kappa_out.sk;
        }
        if constexpr (CMD) {                                     // dq = -gain * pinv(X) (kappa o err) (experiment.py:300-312)
            double panel[R][N + 1];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < N; ++j) panel[r][j] = lds_x[r * N + j][lane];
                panel[r][N] = kap[r] * err[r];
            }
            double sol[N];
            flagged |= alive && lstsq_tall_tuned<M, N, L>(panel, sub, sol);
            if (pc) {                                            // each lane of the pair logs half of the command
                double *po = pc;
#pragma unroll
                for (int u = 0; u < JG; ++u) {
                    const double own = sub ? in_reg(sol[JG + u]) : in_reg(sol[u]);
                    *po = -fp.gain * own;
                    po += A.dqcmd_out.sc;
                }
                pc += A.dqcmd_out.sk;
            }
        }
    }

    if (!valid) return;
    if (sub == 0) {
        if (A.status) A.status[trial] = flagged ? UVS_STATUS_SUSPECT : status;
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
                for (int j = 0; j < N; ++j)
                    *A.p_final.at(trial, 0, ((r * L + sub) * N + l) * N + j) =
                        (r < PV) ? p[r < PV ? r : 0][Sym<N>::at(l, j)] : lds_p[(r >= PV ? r - PV : 0) * NP + Sym<N>::at(l, j)][lane];
    }
}

}  // namespace uvs

namespace uvs {

// Sum over the L lanes lane, lane ^ (64 / L), ... that hold one filter under the blocked mapping; every lane gets the bit-identical total
// (butterfly through the LDS crossbar, ds_bpermute; used once or twice per step).
template <int L>
UVS_DEV double blocked_sum(double v) {
#pragma unroll
    for (int d = 64 / L; d < 64; d *= 2) v += __shfl_xor(v, d, 64);
    return v;
}

// Estimator alone (no control law), L lanes per filter, whole state in registers, two wavefronts per SIMD.  Without the least-squares
// solve the rows of a filter never talk to each other (except the finiteness probe and IMCC-KF's shared weight), so four lanes per
// filter cost no redundant arithmetic, the per-lane state (2 covariance blocks + 2 rows of X = 54 doubles at (8,6)) fits a 256-register
// budget, and a second wavefront on the SIMD issues its arithmetic under this one's stores: the replay becomes HBM-bound.
// EOUT (err stream wanted) is compile-time like XOUT: the prefetched inputs are waited for with an in-order vmcnt that must let every
// store issued after them stay in flight, and the compiler can only count stores it knows will be issued.
template <int M, int N, int L, int METHOD, bool XOUT, bool EOUT>
__global__ __launch_bounds__(64, 2) void replay_rows_kernel(const ReplayArgs A) {
    static_assert((L == 2 || L == 4) && M % L == 0, "rows kernel: 2 or 4 lanes per filter");
    constexpr int R = M / L, NP = Sym<N>::NP, TPW = 64 / L;
    // Lane -> (trial, sub) is blocked, not interleaved: lanes [sub * TPW, (sub + 1) * TPW) hold row group `sub` of TPW consecutive
    // trials, so the four lanes of every quad store to adjacent addresses.  With the interleaved mapping of the least-squares kernels
    // (partner lanes adjacent, for DPP) a quad scatters over L component rows and the same stores cost 3x as much here.
    const unsigned lane = threadIdx.x;
    const int sub = (int)(lane / TPW);
    const long long wave_first = (long long)blockIdx.x * TPW;
    const unsigned tl = lane % TPW;
    const bool valid = wave_first + tl < A.T;
    const long long trial = valid ? wave_first + tl : A.T - 1;     // padding lanes shadow the last trial
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;

    const double *pf = A.f.at(trial, 1, sub);
    const double *pd = A.dq.at(trial, 1, 0);
    double *px = (XOUT && A.x_out.p) ? A.x_out.at(trial, 0, sub * N) : nullptr;
    double *pe = (EOUT && A.err_out.p) ? A.err_out.at(trial, 0, sub) : nullptr;
    double *pk = A.kappa_out.p ? A.kappa_out.at(trial, 0, sub) : nullptr;
    const bool on_kappa = A.kappa_out.p != nullptr;

    double f_prev[R], des[R], f_next[R], h_next[N], x[R][N], p[R][NP];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        des[r] = pick_sub<L>(&fp.desired[r * L], sub);
        f_prev[r] = *A.f.at(trial, 0, r * L + sub);
        f_next[r] = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) x[r][j] = *A.x0.at(trial, 0, (r * L + sub) * N + j);
#pragma unroll
        for (int l = 0; l < N; ++l)
#pragma unroll
            for (int j = l; j < N; ++j) p[r][Sym<N>::at(l, j)] = (l == j) ? 1.0 : 0.0;      // P = I (experiment.py:73)
    }
#pragma unroll
    for (int j = 0; j < N; ++j) h_next[j] = 0.0;                    // first_run: H = 0 (experiment.py:183-185)
    if (K > 0) {
        const double *pr = pf;
#pragma unroll
        for (int r = 0; r < R; ++r) { f_next[r] = *pr; pr += L * A.f.sc; }
        pf += A.f.sk;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0), see replay_tuned_kernel

    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true, flagged = false;                          // flagged (MCKF): a step needs more than the first fixed-point pass
    for (int k = 0; k < K; ++k) {
        double f[R], dq[N];
#pragma unroll
        for (int r = 0; r < R; ++r) f[r] = f_next[r];
#pragma unroll
        for (int j = 0; j < N; ++j) dq[j] = h_next[j];
        if (k + 1 < K) {
            const double *pr = pf;
#pragma unroll
            for (int r = 0; r < R; ++r) { f_next[r] = *pr; pr += L * A.f.sc; }
            const double *pj = pd;
#pragma unroll
            for (int j = 0; j < N; ++j) { h_next[j] = *pj; pj += A.dq.sc; }
            pf += A.f.sk;
            pd += A.dq.sk;
        }
        const double sigma = bandwidth(fp, k);
        const double neg_half_inv_s2 = -0.5 * fast_rcp(sigma * sigma);
        double c_shared = 1.0;
        if constexpr (METHOD == UVS_METHOD_IMCCKF) {
            double ss = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double pred = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) pred = fma(x[r][j], dq[j], pred);
                const double nu = (f[r] - f_prev[r]) - pred;
                ss = fma(nu, nu, ss);
            }
            c_shared = exp_nonpos(blocked_sum<L>(ss) * neg_half_inv_s2);
        }
        double kap[R], err[R];
        double chk = 0.0;
        FpiProbe fpi;
        if constexpr (METHOD == UVS_METHOD_MCKF) {
            mckf_underflow_prepass<R>(fpi, [&](int r) {
                double pred = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) pred = fma(x[r][j], dq[j], pred);
                const double nu = (f[r] - f_prev[r]) - pred;
                return (nu * nu) * neg_half_inv_s2;
            });
            fpi.skip = blocked_sum<L>(fpi.skip ? 1.0 : 0.0) != 0.0;
            fpi.unsure = blocked_sum<L>(fpi.unsure ? 1.0 : 0.0) != 0.0;
        }
        double *pxr = px;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double zi = f[r] - f_prev[r];
            f_prev[r] = f[r];
            err[r] = f[r] - des[r];
            rmckf_row<N, METHOD>(x[r], p[r], dq, zi, neg_half_inv_s2, c_shared, fp.reg, kap[r], chk, fpi);
            if constexpr (XOUT) {
                double *pcx = pxr;
#pragma unroll
                for (int j = 0; j < N; ++j) { *pcx = x[r][j]; pcx += A.x_out.sc; }
                pxr += L * N * A.x_out.sc;
            }
        }
        if constexpr (XOUT) px += A.x_out.sk;
        chk = blocked_sum<L>(chk);
        if (alive && !(chk == 0.0)) {
            alive = false;
            status = UVS_STATUS_FAIL;
            k_done = k;
        }
        if constexpr (METHOD == UVS_METHOD_MCKF) {
            fpi.num = blocked_sum<L>(fpi.num);
            fpi.den = blocked_sum<L>(fpi.den);
            flagged |= alive && fpi_needs_more(fpi, fp);
        }
        // no early exit when every trial of the wavefront has failed: a path that skips the err stores would make the compiler's
        // in-order vmcnt for the prefetched inputs count only the X stores and wait for the rest, every step
        if constexpr (EOUT) {
            double *po = pe;
#pragma unroll
            for (int r = 0; r < R; ++r) { *po = err[r]; po += L * A.err_out.sc; }
            pe += A.err_out.sk;
        }
        if (on_kappa) {
            double *po = pk;
#pragma unroll
            for (int r = 0; r < R; ++r) { *po = kap[r]; po += L * A.kappa_out.sc; }
            pk += A.kappa_out.sk;
        }
    }

    if (!valid) return;
    if (sub == 0) {
        if (A.status) A.status[trial] = flagged ? UVS_STATUS_SUSPECT : status;
        if (A.k_done) A.k_done[trial] = k_done;
    }
    if (A.x_final.on()) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) *A.x_final.at(trial, 0, (r * L + sub) * N + j) = x[r][j];
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

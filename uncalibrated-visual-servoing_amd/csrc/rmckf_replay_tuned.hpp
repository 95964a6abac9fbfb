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
            fpi.poison = (pair_sum<L>(fpi.poison ? 1.0 : 0.0) != 0.0) && fp.fpi_epoch_max > 1;    // (a cap of one pass goes to the careful pass anyway)
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
        if constexpr (METHOD == UVS_METHOD_MCKF) chk = mckf_poisoned(fpi, chk);
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

// One trial per lane, everything in registers: sol = pinv(J) y by the normal equations with one refinement step (a control wavefront of
// the replay carries nothing but the panel).  Split in two halves so that a workgroup barrier can sit between them.
template <int M, int N>
UVS_DEV void normal_eq_gram(const double (&x)[M][N], const double (&y)[M], double (&G)[Sym<N>::NP], double (&b)[N]) {
#pragma unroll
    for (int l = 0; l < N; ++l) {
#pragma unroll
        for (int j = l; j < N; ++j) {
            double acc = x[0][l] * x[0][j];
#pragma unroll
            for (int i = 1; i < M; ++i) acc = fma(x[i][l], x[i][j], acc);
            G[Sym<N>::at(l, j)] = acc;
        }
        double acc = x[0][l] * y[0];
#pragma unroll
        for (int i = 1; i < M; ++i) acc = fma(x[i][l], y[i], acc);
        b[l] = acc;
    }
}
template <int M, int N>
UVS_DEV bool normal_eq_finish(const double (&x)[M][N], const double (&y)[M], double (&G)[Sym<N>::NP], double (&b)[N], double (&sol)[N]) {
    double rs[N], c[N];
    bool suspect = chol_factor<N>(G, rs);
    chol_solve_inplace<N>(G, rs, b);                             // s0
#pragma unroll
    for (int j = 0; j < N; ++j) c[j] = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {                                // c = J^T (y - J s0)
        double ri = y[i];
#pragma unroll
        for (int j = 0; j < N; ++j) ri = fma(-x[i][j], b[j], ri);
#pragma unroll
        for (int j = 0; j < N; ++j) c[j] = fma(x[i][j], ri, c[j]);
    }
    chol_solve_inplace<N>(G, rs, c);
    double s_max = 0.0, c_max = 0.0;                             // refinement watch (rmckf_wide.hpp): a correction of 2^-20 of the solution marks the trial
#pragma unroll
    for (int j = 0; j < N; ++j) { s_max = fmax(s_max, fabs(b[j])); c_max = fmax(c_max, fabs(c[j])); }
    suspect |= (unsigned)__double2hiint(c_max) + kRefineGate >= (unsigned)__double2hiint(s_max) && c_max > 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) sol[j] = b[j] + c[j];
    return suspect;
}

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
//
// BYWAVE (KF and RMCKF, whose rows share nothing but the finiteness verdict): the L row groups of a filter are the L WAVEFRONTS of a
// workgroup instead of L lane groups of one wavefront -- lane = trial, wavefront = row group, 64 consecutive trials per workgroup.
// Same registers, same arithmetic, no cross-lane traffic at all, and every load and store of a wavefront covers 64 consecutive trials:
// 512 contiguous bytes in the trial-fastest layout instead of four 128-byte segments.  The verdicts meet in LDS once, after the loop.
//
// REC (lane groups only): the X and err streams are per-trial records ([step][trial][component]: comp_stride 1, trial_stride M N resp. M).
// The TPW trials of a wavefront are then ONE contiguous block per step (TPW M N doubles of X, TPW M of err): the finished rows are
// transposed through a wavefront-private LDS buffer and leave as whole 1 KB stores (64 lanes x 16 B), issued from the hook points of the
// next step's first row.
//
// CW = 2 (BYWAVE only; the commanded dq is wanted): two more wavefronts per workgroup run the control law dq = -gain pinv(X_k)(kappa o err)
// with one trial per lane.  In a replay nothing waits for that result -- the next regressor comes from the recorded stream -- so the
// estimator wavefronts only drop X_k, kappa o err and their finiteness probe into a double-buffered LDS block and meet the control
// wavefronts at one barrier per step; control wavefront c takes the steps k = c (mod 2) and has two barrier intervals for each (Gram
// matrix in the first, Cholesky + solve + refinement + stores in the second).  Normal equations: ill-conditioned Jacobians are marked
// for the careful second pass exactly as in the closed-loop kernels.
template <int M, int N, int L, int METHOD, bool XOUT, bool EOUT, bool BYWAVE = false, bool REC = false, int CW = 0>
__global__ __launch_bounds__(BYWAVE ? 64 * (L + CW) : 64, 2) void replay_rows_kernel(const ReplayArgs A) {
    static_assert((L == 2 || L == 4) && M % L == 0, "rows kernel: 2 or 4 lanes per filter");
    static_assert(CW == 0 || (CW == 2 && BYWAVE), "control wavefronts: with the row groups in wavefronts");
    static_assert(!(REC && BYWAVE) && (!REC || (XOUT && EOUT)), "record path: lane groups, X and err both wanted");
    static_assert(!BYWAVE || METHOD == UVS_METHOD_GMCKF || METHOD == UVS_METHOD_KF, "row groups in separate wavefronts: independent rows only");
    constexpr int R = M / L, NP = Sym<N>::NP, TPW = BYWAVE ? 64 : 64 / L;
    // Lane -> (trial, sub) is blocked, not interleaved: lanes [sub * TPW, (sub + 1) * TPW) hold row group `sub` of TPW consecutive
    // trials, so the four lanes of every quad store to adjacent addresses.  With the interleaved mapping of the least-squares kernels
    // (partner lanes adjacent, for DPP) a quad scatters over L component rows and the same stores cost 3x as much here.
    const unsigned lane = threadIdx.x;
    const int sub = BYWAVE ? __builtin_amdgcn_readfirstlane((int)(lane >> 6)) : (int)(lane / TPW);
    const long long wave_first = (long long)blockIdx.x * TPW;
    const unsigned tl = lane % TPW;
    const bool valid = wave_first + tl < A.T;
    const long long trial = valid ? wave_first + tl : A.T - 1;     // padding lanes shadow the last trial
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;

    constexpr int CV = M * N + M;                                  // values an estimator step hands to the control law: X_k, kappa o err
    __shared__ double lcb[CW ? 2 : 1][CW ? CV : 1][64];
    __shared__ double lcc[CW ? 2 : 1][CW ? L : 1][64];             // finiteness probes of the row groups
    __shared__ int lcflag[64];
    if constexpr (CW > 0) {
        if (lane < 64) lcflag[lane] = 0;
        if (sub >= L) {                                            // ---- control wavefront c = sub - L: steps k = c (mod 2)
            const int c = sub - L;
            double xs[M][N], ys[M], G[NP], b[N];
            bool ok = false;
            __syncthreads();                                       // lcflag cleared
            for (int k = 0; k <= K; ++k) {
                if (k < K) __syncthreads();                        // barrier k: the block of step k is complete
                if (k < K && (k & 1) == c) {
#pragma unroll
                    for (int i = 0; i < M; ++i) {
#pragma unroll
                        for (int j = 0; j < N; ++j) xs[i][j] = lcb[k & 1][i * N + j][tl];
                        ys[i] = lcb[k & 1][M * N + i][tl];
                    }
                    double probe = 0.0;
#pragma unroll
                    for (int g = 0; g < L; ++g) probe += lcc[k & 1][g][tl];
                    ok = probe == 0.0;
                    normal_eq_gram<M, N>(xs, ys, G, b);
                } else if (k >= 1 && ((k - 1) & 1) == c) {
                    double sol[N];
                    const bool suspect = normal_eq_finish<M, N>(xs, ys, G, b, sol);
                    if (ok && suspect) lcflag[tl] = 1;             // ill-conditioned Jacobian: the careful second pass redoes this trial
                    double *po = A.dqcmd_out.at(trial, k - 1, 0);
#pragma unroll
                    for (int j = 0; j < N; ++j) po[j * A.dqcmd_out.sc] = -fp.gain * sol[j];      // experiment.py:312
                }
            }
            __syncthreads();                                       // the estimator wavefronts' verdict exchange
            return;
        }
        __syncthreads();                                           // lcflag cleared
    }

    const double *pf = A.f.at(trial, 1, sub);
    const double *pd = A.dq.at(trial, 1, 0);
    double *px = (XOUT && A.x_out.p) ? (REC ? A.x_out.at(wave_first, 0, 2 * lane) : A.x_out.at(trial, 0, sub * N)) : nullptr;
    double *pe = (EOUT && A.err_out.p) ? (REC ? A.err_out.at(wave_first, 0, 2 * lane) : A.err_out.at(trial, 0, sub)) : nullptr;
    double *pk = A.kappa_out.p ? A.kappa_out.at(trial, 0, sub) : nullptr;
    const bool on_kappa = A.kappa_out.p != nullptr;
    // record path: LDS copies of the wavefront's block, records padded by one double against bank conflicts; src_x[i]: where the pair of
    // doubles this lane stores with the i-th 1 KB store lives (a pair never straddles two records: M N and M are even)
    constexpr int XREC = M * N, XRECP = XREC + 1, EREC = M, ERECP = EREC + 1, XS = TPW * XREC / 128, ES = TPW * EREC / 128;
    static_assert(!REC || ((TPW * XREC) % 128 == 0 && (TPW * EREC) % 128 == 0 && XS + ES <= N + 1), "record path: whole 1 KB stores, one per hook point of a row");
    __shared__ double ltx[REC ? TPW * XRECP : 1], lte[REC ? TPW * ERECP : 1];
    int src_x[REC ? XS : 1], src_e[REC ? ES : 1];
    if constexpr (REC) {
#pragma unroll
        for (int i = 0; i < XS; ++i) { const int d = 128 * i + 2 * (int)lane; src_x[i] = (d / XREC) * XRECP + d % XREC; }
#pragma unroll
        for (int i = 0; i < ES; ++i) { const int d = 128 * i + 2 * (int)lane; src_e[i] = (d / EREC) * ERECP + d % EREC; }
    }

    double f_prev[R], des[R], f_next[R], h_next[N], x[R][N], p[R][NP];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if constexpr (BYWAVE) des[r] = fp.desired[r * L + sub];
        else des[r] = pick_sub<L>(&fp.desired[r * L], sub);
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
    // The stores of a step are software-pipelined behind the arithmetic (rmckf_row's hook points): while row r is updated, the X row and the
    // error of row r - 1 -- finished, and untouched until the next step -- go out one store at a time; row R - 1 of a step goes out
    // during row 0 of the next step, and after the last step on its own.  `step` is the loop body, FIRST = nothing pending yet (the first
    // step is peeled instead of branching around the stores: the compiler's in-order vmcnt for the prefetched inputs must see every store).
    double err_last = 0.0;
    auto step = [&](int k, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
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
            fpi.poison = (blocked_sum<L>(fpi.poison ? 1.0 : 0.0) != 0.0) && fp.fpi_epoch_max > 1;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double zi = f[r] - f_prev[r];
            f_prev[r] = f[r];
            err[r] = f[r] - des[r];
            // pending: row r - 1 of this step, or row R - 1 of the previous one
            const int rr = (r + R - 1) % R;
            const double (&xs)[N] = x[rr];
            const double es = (r == 0) ? err_last : err[rr];
            double *pxs = px ? px + (long long)rr * L * N * A.x_out.sc - (r == 0 ? A.x_out.sk : 0) : nullptr;
            double *pes = pe ? pe + (long long)rr * L * A.err_out.sc - (r == 0 ? A.err_out.sk : 0) : nullptr;
            auto hook = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (REC) {
                    if (r == 0) {                                  // the previous step's block: LDS -> one 1 KB store per hook point
                        if constexpr (!FIRST) {
                            if constexpr (i < XS) {
                                const double *src = &ltx[src_x[i]];
                                double2 v;
                                v.x = src[0];
                                v.y = src[1];
                                *reinterpret_cast<double2 *>(px + 128 * i - A.x_out.sk) = v;
                            } else if constexpr (i < XS + ES) {
                                const double *src = &lte[src_e[i - XS]];
                                double2 v;
                                v.x = src[0];
                                v.y = src[1];
                                *reinterpret_cast<double2 *>(pe + 128 * (i - XS) - A.err_out.sk) = v;
                            }
                        }
                    } else {                                       // row r - 1 of this step into the LDS block
                        if constexpr (i < N) ltx[tl * XRECP + (rr * L + sub) * N + i] = xs[i];
                        else lte[tl * ERECP + rr * L + sub] = es;
                    }
                } else {
                    if constexpr (FIRST) { if (r == 0) return; }
                    if constexpr (i < N) { if constexpr (XOUT) pxs[i * A.x_out.sc] = xs[i]; }
                    else { if constexpr (EOUT) *pes = es; }
                }
            };
            rmckf_row<N, METHOD>(x[r], p[r], dq, zi, neg_half_inv_s2, c_shared, fp.reg, kap[r], chk, fpi, hook);
        }
        err_last = err[R - 1];
        if constexpr (CW > 0) {                                    // hand X_k, kappa o err and the probe to the control wavefronts
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < N; ++j) lcb[k & 1][(r * L + sub) * N + j][tl] = x[r][j];
                lcb[k & 1][M * N + r * L + sub][tl] = kap[r] * err[r];
            }
            lcc[k & 1][sub][tl] = chk;
            __syncthreads();                                       // barrier k
        }
        if constexpr (REC) {                                       // the last row of this step into the LDS block
#pragma unroll
            for (int j = 0; j < N; ++j) ltx[tl * XRECP + ((R - 1) * L + sub) * N + j] = x[R - 1][j];
            lte[tl * ERECP + (R - 1) * L + sub] = err_last;
        }
        if constexpr (XOUT) px += A.x_out.sk;
        if constexpr (EOUT) pe += A.err_out.sk;
        if constexpr (!BYWAVE) chk = blocked_sum<L>(chk);
        if constexpr (METHOD == UVS_METHOD_MCKF) chk = mckf_poisoned(fpi, chk);
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
        if (on_kappa) {
            double *po = pk;
#pragma unroll
            for (int r = 0; r < R; ++r) { *po = kap[r]; po += L * A.kappa_out.sc; }
            pk += A.kappa_out.sk;
        }
    };
    if (K > 0) {
        step(0, std::true_type{});
        for (int k = 1; k < K; ++k) step(k, std::false_type{});
        if constexpr (REC) {                                       // the block of the last step
#pragma unroll
            for (int i = 0; i < XS; ++i) {
                const double *src = &ltx[src_x[i]];
                double2 v;
                v.x = src[0];
                v.y = src[1];
                *reinterpret_cast<double2 *>(px + 128 * i - A.x_out.sk) = v;
            }
#pragma unroll
            for (int i = 0; i < ES; ++i) {
                const double *src = &lte[src_e[i]];
                double2 v;
                v.x = src[0];
                v.y = src[1];
                *reinterpret_cast<double2 *>(pe + 128 * i - A.err_out.sk) = v;
            }
        } else {                                                   // row R - 1 of the last step
            if constexpr (XOUT) {
                double *po = px + (long long)(R - 1) * L * N * A.x_out.sc - A.x_out.sk;
#pragma unroll
                for (int j = 0; j < N; ++j) po[j * A.x_out.sc] = x[R - 1][j];
            }
            if constexpr (EOUT) *(pe + (long long)(R - 1) * L * A.err_out.sc - A.err_out.sk) = err_last;
        }
    }

    if constexpr (BYWAVE) {                                       // a trial fails at the first step at which any of its row groups turned non-finite
        __shared__ int lk[L][64];
        lk[sub][tl] = k_done;
        __syncthreads();
#pragma unroll
        for (int g = 0; g < L; ++g) k_done = lk[g][tl] < k_done ? lk[g][tl] : k_done;
        status = (k_done < K) ? UVS_STATUS_FAIL : UVS_STATUS_SUCCESS;
        if constexpr (CW > 0) flagged = lcflag[tl] != 0;
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

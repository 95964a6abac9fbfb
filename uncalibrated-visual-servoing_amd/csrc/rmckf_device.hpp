// Device-side building blocks of the batched RMCKF estimator (gfx950 / CDNA4, wave64, fp64).
//
// Work decomposition.  A *filter* is one Monte-Carlo trial's estimator: m rows x_i (n doubles each)
// and m covariance blocks P_i (n x n, symmetric, stored packed), all rows sharing the regressor
// h = dq (experiment.py:188: H = kron(I_m, dq^T) makes the reference's mn x mn P exactly block
// diagonal).  L lanes of a wavefront cooperate on one filter, each owning
// R = m / L consecutive rows in registers; 64 / L filters ride in one wavefront.  L = 1 is
// "one filter per lane": no cross-lane traffic at all.  L > 1 trades registers for wavefront
// shuffles (xor butterflies inside the L-lane group) in the innovation norm, the least-squares
// control law and the final statistics.
//
// Everything here is templated on (M, N, L) and fully unrolled so that every array lives in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "uvs_rmckf.h"
#include "rmckf_math.hpp"

namespace uvs {

#ifndef UVS_DEV
#define UVS_DEV __device__ __forceinline__
#endif

// ---------------------------------------------------------------- cross-lane helpers
// The L lanes of a filter are consecutive and L divides 64.  Up to 16 lanes form (part of) one DPP row, so reductions and
// broadcasts are DPP moves (VALU, no LDS): butterfly stages quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror,
// row_mirror; beyond a row the last stages fall back to ds_bpermute (__shfl_xor).  Verified on gfx950 (tools/ubench/dpp_test).
template <int CTRL>
UVS_DEV int dpp_mov32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
UVS_DEV double dpp_mov64(double v) {
    return __hiloint2double(dpp_mov32<CTRL>(__double2hiint(v)), dpp_mov32<CTRL>(__double2loint(v)));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppNewBcast = 0x150;

// Sum over the L lanes of a group; every lane receives the bit-identical total (both partners of a stage add the same pair).
template <int L>
UVS_DEV double group_sum(double v) {
    if constexpr (L >= 2) v += dpp_mov64<kDppXor1>(v);
    if constexpr (L >= 4) v += dpp_mov64<kDppXor2>(v);
    if constexpr (L >= 8) v += dpp_mov64<kDppHalfMirror>(v);
    if constexpr (L >= 16) v += dpp_mov64<kDppMirror>(v);
    if constexpr (L >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (L >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
// Largest of the group's non-negative values (NaN-free by construction: callers feed fmax results), on every lane.
template <int L>
UVS_DEV double group_max(double v) {
    if constexpr (L >= 2) v = fmax(v, dpp_mov64<kDppXor1>(v));
    if constexpr (L >= 4) v = fmax(v, dpp_mov64<kDppXor2>(v));
    if constexpr (L >= 8) v = fmax(v, dpp_mov64<kDppHalfMirror>(v));
    if constexpr (L >= 16) v = fmax(v, dpp_mov64<kDppMirror>(v));
    if constexpr (L >= 32) v = fmax(v, __shfl_xor(v, 16, 64));
    if constexpr (L >= 64) v = fmax(v, __shfl_xor(v, 32, 64));
    return v;
}
template <int L>
UVS_DEV int group_or(int v) {
    if constexpr (L >= 2) v |= dpp_mov32<kDppXor1>(v);
    if constexpr (L >= 4) v |= dpp_mov32<kDppXor2>(v);
    if constexpr (L >= 8) v |= dpp_mov32<kDppHalfMirror>(v);
    if constexpr (L >= 16) v |= dpp_mov32<kDppMirror>(v);
    if constexpr (L >= 32) v |= __shfl_xor(v, 16, 64);
    if constexpr (L >= 64) v |= __shfl_xor(v, 32, 64);
    return v;
}
// Broadcast of group lane OWNER to the whole group (L <= 16): quad_perm inside a quad, row_newbcast inside a row
// (with bank masks when two groups of 8 share the row).
template <int L, int OWNER>
UVS_DEV int group_bcast32(int v) {
    if constexpr (L == 2) return dpp_mov32<OWNER | (OWNER << 2) | ((2 + OWNER) << 4) | ((2 + OWNER) << 6)>(v);
    else if constexpr (L == 4) return dpp_mov32<OWNER * 0x55>(v);
    else if constexpr (L == 8) {
        int r = __builtin_amdgcn_update_dpp(0, v, kDppNewBcast + OWNER, 0xf, 0x3, false);
        return __builtin_amdgcn_update_dpp(r, v, kDppNewBcast + 8 + OWNER, 0xf, 0xC, false);
    } else return dpp_mov32<kDppNewBcast + OWNER>(v);
}
template <int L, int OWNER>
UVS_DEV double group_bcast(double v) {
    return __hiloint2double(group_bcast32<L, OWNER>(__double2hiint(v)), group_bcast32<L, OWNER>(__double2loint(v)));
}
// Value held by the lane whose group-relative index is `owner` (a compile-time constant after unrolling; others pass anything).
template <int L>
UVS_DEV double group_pick(double v, int sub, int owner) {
    if constexpr (L == 1) return v;
    else if constexpr (L > 16) return group_sum<L>(sub == owner ? v : 0.0);
    else {
        switch (owner & (L - 1)) {
#define UVS_CASE(O) case O: return group_bcast<L, (O < L ? O : 0)>(v);
            UVS_CASE(0) UVS_CASE(1) UVS_CASE(2) UVS_CASE(3) UVS_CASE(4) UVS_CASE(5) UVS_CASE(6) UVS_CASE(7)
            UVS_CASE(8) UVS_CASE(9) UVS_CASE(10) UVS_CASE(11) UVS_CASE(12) UVS_CASE(13) UVS_CASE(14) UVS_CASE(15)
#undef UVS_CASE
            default: return v;
        }
    }
}

UVS_DEV bool finite64(double v) { return (__double_as_longlong(v) & 0x7ff0000000000000LL) != 0x7ff0000000000000LL; }

// ---------------------------------------------------------------- strided views
struct View {
    double *p;
    long long st, sk, sc;
    UVS_DEV bool on() const { return p != nullptr; }
    UVS_DEV double *at(long long t, long long k, long long c) const { return p + t * st + k * sk + c * sc; }
};
static inline View to_view(const uvs_view &v) { return View{v.base, v.trial_stride, v.step_stride, v.comp_stride}; }

// ---------------------------------------------------------------- kernel argument blocks (passed by value)
constexpr int kMaxSegments = 16;
struct ClosedArgs {
    uvs_filter_params fp;
    uvs_plant plant;
    long long T;
    View q_start, noise, x0, x_out, err_out, q_out, f_out, dq_out, x_final, p_final;
    double *stats;
    int *status, *k_done;
    // Segmented trials (tuned two-lane MCKF kernel, uvs_rmckf_closed_loop_ws_f64): n_seg > 1 cuts the K steps of a trial chunk into n_seg work
    // items; the filter state crosses from one item to the next through ws_state, ws_flags[chunk] counts the chunk's finished segments.
    double *ws_state = nullptr;
    int *ws_flags = nullptr;
    int n_seg = 1;
    int seg_first[kMaxSegments + 1] = {};       // segment s covers steps [seg_first[s], seg_first[s + 1])
};
// Doubles per lane that one trial-chunk state occupies in the workspace (an upper bound over the kernel variants of a shape, so that the
// C ABI can size the workspace without knowing which one runs): joints + their sines / cosines, command, previous features, clock,
// covariance blocks, X, the twelve statistics accumulators, one word of flags.
constexpr int seg_state_doubles(int M, int N, int L) {
    return 3 * N + N + (M / L) + 1 + (M / L) * (N * (N + 1) / 2) + (M / L) * N + 3 * (M / L) + 1;
}
// x ~1 us of s_sleep 32: after ~65 ms without its predecessor (a segment lasts < 1 ms) a later segment recomputes the trial from step 0 instead.
// Short on purpose: should in-order dispatch ever not hold (another dispatcher policy, a debugger, a partitioned GPU) the launch degrades to
// recomputation within a watchdog's patience instead of sitting seconds per segment (round 4: 1 << 22, ~4 s).
constexpr int kSegSpinMax = 1 << 16;

struct ReplayArgs {
    uvs_filter_params fp;
    long long T;
    View f, dq, x0, x_out, err_out, kappa_out, dqcmd_out, x_final, p_final;
    int *status, *k_done;
};

struct StepArgs {
    uvs_filter_params fp;
    long long T;
    double *X, *P;
    const double *f, *f_old, *dq_prev;
    int first, k;
    double *dq_out, *err_out, *kappa_out;
    int *status;
};

// ---------------------------------------------------------------- packed symmetric n x n
template <int N>
struct Sym {
    static constexpr int NP = N * (N + 1) / 2;
    __host__ __device__ static constexpr int at(int l, int j) {
        return l <= j ? l * N - l * (l - 1) / 2 + (j - l) : j * N - j * (j - 1) / 2 + (l - j);
    }
};

// ---------------------------------------------------------------- estimator rows
// sigma_k (experiment.py:267-271)
UVS_DEV double bandwidth(const uvs_filter_params &fp, int k) {
    return fp.annealing ? fp.kernel_bw + fp.anneal_span * (1.0 - (double)k / (double)fp.k_max) : fp.kernel_bw;
}
// utils.py:171-172 with numpy's evaluation order: ((-0.5 * e*e) / (bw*bw))
UVS_DEV double gaussian_kernel(double e, double bw) { return exp((-0.5 * (e * e)) / (bw * bw)); }

template <int M, int N, int L>
struct Rows {
    static constexpr int R = M / L;
    static constexpr int NP = Sym<N>::NP;
    static_assert(M % L == 0, "lanes per filter must divide m");
    double x[R][N];
    double p[R][NP];

    UVS_DEV void init_cov() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = l; j < N; ++j) p[r][Sym<N>::at(l, j)] = (l == j) ? 1.0 : 0.0;   // P = I (experiment.py:73)
        }
    }

    // One predict + correct for the R rows of this lane (experiment.py:166-297 in row form):
    //   P_i += I; nu_i = z_i - x_i.h; weight; g = P_i h; a = h.g; gamma; x_i += gamma g nu_i;
    //   Joseph with R = 1 collapses to the symmetric rank-1 downdate P_i -= gamma (2 - gamma (a + 1)) g g^T.
    // kap[] receives the correntropy weights kappa_i that the control law re-uses (experiment.py:308).
    // METHOD_T != 0 fixes the estimator at compile time (the other paths are not even compiled into that kernel, which keeps
    // its register allocation at what the estimator needs); METHOD_T == 0 reads fp.method at run time.
    template <int METHOD_T = 0>
    UVS_DEV void update(const uvs_filter_params &fp_in, const double (&z)[R], const double (&h)[N], double sigma, double (&kap)[R]) {
        struct { int method; double reg, fpi_threshold; int fpi_epoch_max; } fp{METHOD_T ? METHOD_T : fp_in.method, fp_in.reg, fp_in.fpi_threshold, fp_in.fpi_epoch_max};
        if constexpr (METHOD_T == 0 || METHOD_T == UVS_METHOD_MCKF) {
            if (fp.method == UVS_METHOD_MCKF) {
                update_mckf(fp_in, z, h, sigma, kap);
                return;
            }
        }
        double nu[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) acc = fma(x[r][j], h[j], acc);
            nu[r] = z[r] - acc;                                                  // innovation (experiment.py:274)
        }
        double c_shared = 1.0;
        if (fp.method == UVS_METHOD_IMCCKF) {                                    // experiment.py:258-261
            double ss = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) ss = fma(nu[r], nu[r], ss);
            c_shared = gaussian_kernel(sqrt(group_sum<L>(ss)), sigma);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double g[N];
#pragma unroll
            for (int l = 0; l < N; ++l) p[r][Sym<N>::at(l, l)] += 1.0;           // predict: P + Q, Q = I (experiment.py:167)
#pragma unroll
            for (int l = 0; l < N; ++l) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) acc = fma(p[r][Sym<N>::at(l, j)], h[j], acc);
                g[l] = acc;
            }
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) a = fma(h[l], g[l], a);
            double gamma;
            if (fp.method == UVS_METHOD_GMCKF) {                                 // experiment.py:276-286
                kap[r] = gaussian_kernel(nu[r], sigma);
                const double dd = kap[r] + fp.reg;                               // 1 / (a + 1/dd) = dd / (a dd + 1)
                gamma = dd * fast_rcp(fma(a, dd, 1.0));
            } else if (fp.method == UVS_METHOD_IMCCKF) {                         // experiment.py:262-264
                kap[r] = 1.0;
                gamma = c_shared * fast_rcp(fma(c_shared, a, 1.0));
            } else {                                                             // KF, experiment.py:192
                kap[r] = 1.0;
                gamma = fast_rcp(a + 1.0);
            }
            const double step = gamma * nu[r];
            const double beta = gamma * (2.0 - gamma * (a + 1.0));
#pragma unroll
            for (int l = 0; l < N; ++l) x[r][l] = fma(g[l], step, x[r][l]);      // X + K (Z - H X) (experiment.py:291)
#pragma unroll
            for (int l = 0; l < N; ++l) {
                const double w = beta * g[l];
#pragma unroll
                for (int j = l; j < N; ++j) p[r][Sym<N>::at(l, j)] = fma(-w, g[j], p[r][Sym<N>::at(l, j)]);
            }
        }
    }

    // Fixed-point MCKF (experiment.py:194-250) row by row.  B = blkdiag(chol(P), chol(R)) is block diagonal, hence
    //   D - W X_c = [ L_i^-1 (x_i - xc_i) ;  z_i - h.xc_i ],   P_hat_i = L_i Cx_i^-1 L_i^T,   R_hat_i = 1 / Cy_i,
    //   k_i = P_hat_i h / (h.P_hat_i h + R_hat_i),   xc_i <- x_i + k_i (z_i - h.x_i)              (the prior innovation, :242)
    // iterated until ||Xc - Xc_old|| / ||Xc_old|| <= fpi_threshold over ALL rows (:244).  A zero in Cy makes inv(Cy) raise
    // and the epoch cap is reached -> the whole correction (X and P) is skipped (:231-236, :246-250).  The first pass has
    // Xc = X, so Cx = I and P_hat = P: no factorisation is needed unless a second pass is.  Joseph (:297) with a gain that is
    // not gamma * P h is the rank-2 form P - k g^T - g k^T + (h.g + 1) k k^T with g = P h.
    UVS_DEV void update_mckf(const uvs_filter_params &fp, const double (&z)[R], const double (&h)[N], double sigma, double (&kap)[R]) {
        double nu0[R], gp[R][N], ap[R], xc[R][N], kk[R][N];
        int bad = 0;
        double num = 0.0, den = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            kap[r] = 1.0;
#pragma unroll
            for (int l = 0; l < N; ++l) p[r][Sym<N>::at(l, l)] += 1.0;           // predict (experiment.py:167)
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) acc = fma(x[r][j], h[j], acc);
            nu0[r] = z[r] - acc;
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) s = fma(p[r][Sym<N>::at(l, j)], h[j], s);
                gp[r][l] = s;
                a = fma(h[l], s, a);
            }
            ap[r] = a;
            const double cy = gaussian_kernel(nu0[r], sigma);                    // first pass: Xc = X
            bad |= (cy == 0.0);
            // a subnormal weight whose reciprocal overflows (cy <= 2^-1024): inv(Cy) = inf, 0 * inf = NaN in the reference's dense
            // Br @ inv(Cy) @ Br.T (:232) -- the gain is NaN, the state follows and the trial FAILs (see kRcpOverflowsAtOrBelow, rmckf_tuned.hpp)
            const double gain = (cy <= 0x1p-1024) ? __builtin_nan("") : 1.0 / (a + 1.0 / cy);
#pragma unroll
            for (int l = 0; l < N; ++l) {
                kk[r][l] = gp[r][l] * gain;
                xc[r][l] = fma(kk[r][l], nu0[r], x[r][l]);
                const double d = xc[r][l] - x[r][l];
                num = fma(d, d, num);
                den = fma(x[r][l], x[r][l], den);
            }
        }
        bool skip = group_or<L>(bad) != 0;
        double diff = sqrt(group_sum<L>(num)) / sqrt(group_sum<L>(den));
        int it = 1;
        if (it == fp.fpi_epoch_max) skip = true;
        bool more = !skip && (diff > fp.fpi_threshold) && it < fp.fpi_epoch_max;  // NaN diff ends the loop like the reference's while
        if (__any(more)) {
            double Lc[R][Sym<N>::NP];                                            // lower Cholesky factors of the predicted blocks, packed
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double dsum = p[r][Sym<N>::at(j, j)];
#pragma unroll
                    for (int k2 = 0; k2 < j; ++k2) dsum = fma(-Lc[r][Sym<N>::at(k2, j)], Lc[r][Sym<N>::at(k2, j)], dsum);
                    const double ljj = sqrt(dsum);
                    Lc[r][Sym<N>::at(j, j)] = ljj;
#pragma unroll
                    for (int i = j + 1; i < N; ++i) {
                        double s = p[r][Sym<N>::at(j, i)];
#pragma unroll
                        for (int k2 = 0; k2 < j; ++k2) s = fma(-Lc[r][Sym<N>::at(k2, i)], Lc[r][Sym<N>::at(k2, j)], s);
                        Lc[r][Sym<N>::at(j, i)] = s / ljj;                       // L[i][j], stored at packed (j, i)
                    }
                }
            }
            while (__any(more)) {
                int bad2 = 0;
                double num2 = 0.0, den2 = 0.0;
                double xn[R][N], kn[R][N];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    double ex[N], t[N], g[N];
#pragma unroll
                    for (int i = 0; i < N; ++i) {                                // L ex = x - xc (forward substitution)
                        double s = x[r][i] - xc[r][i];
#pragma unroll
                        for (int k2 = 0; k2 < i; ++k2) s = fma(-Lc[r][Sym<N>::at(k2, i)], ex[k2], s);
                        ex[i] = s / Lc[r][Sym<N>::at(i, i)];
                    }
                    double ez = z[r];
#pragma unroll
                    for (int j = 0; j < N; ++j) ez = fma(-xc[r][j], h[j], ez);
                    const double cy = gaussian_kernel(ez, sigma);
                    bad2 |= (cy == 0.0);
#pragma unroll
                    for (int j = 0; j < N; ++j) {                                // t = Cx^-1 L^T h
                        double s = 0.0;
#pragma unroll
                        for (int i = j; i < N; ++i) s = fma(Lc[r][Sym<N>::at(j, i)], h[i], s);
                        t[j] = s / gaussian_kernel(ex[j], sigma);
                    }
                    double a = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) {                                // g = L t = P_hat h
                        double s = 0.0;
#pragma unroll
                        for (int j = 0; j <= i; ++j) s = fma(Lc[r][Sym<N>::at(j, i)], t[j], s);
                        g[i] = s;
                        a = fma(h[i], s, a);
                    }
                    const double gain = (cy <= 0x1p-1024) ? __builtin_nan("") : 1.0 / (a + 1.0 / cy);
#pragma unroll
                    for (int l = 0; l < N; ++l) {
                        kn[r][l] = g[l] * gain;
                        xn[r][l] = fma(kn[r][l], nu0[r], x[r][l]);
                        const double d = xn[r][l] - xc[r][l];
                        num2 = fma(d, d, num2);
                        den2 = fma(xc[r][l], xc[r][l], den2);
                    }
                }
                const bool hit_zero = group_or<L>(bad2) != 0;
                const double diff2 = sqrt(group_sum<L>(num2)) / sqrt(group_sum<L>(den2));
                if (more) {                                                      // filters that already stopped keep their result
                    if (hit_zero) {
                        skip = true;
                        more = false;
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int l = 0; l < N; ++l) { xc[r][l] = xn[r][l]; kk[r][l] = kn[r][l]; }
                        ++it;
                        if (it == fp.fpi_epoch_max) skip = true;
                        more = !skip && (diff2 > fp.fpi_threshold) && it < fp.fpi_epoch_max;
                    }
                }
            }
        }
        if (!skip) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double c2 = ap[r] + 1.0;
#pragma unroll
                for (int l = 0; l < N; ++l) x[r][l] = xc[r][l];
#pragma unroll
                for (int l = 0; l < N; ++l)
#pragma unroll
                    for (int j = l; j < N; ++j) {
                        double v = p[r][Sym<N>::at(l, j)];
                        v = fma(-kk[r][l], gp[r][j], v);
                        v = fma(-gp[r][l], kk[r][j], v);
                        v = fma(c2 * kk[r][l], kk[r][j], v);
                        p[r][Sym<N>::at(l, j)] = v;
                    }
            }
        }
    }

    UVS_DEV int any_nonfinite() const {
        int bad = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) bad |= !finite64(x[r][j]);
        return group_or<L>(bad);
    }
};

// ---------------------------------------------------------------- numpy.linalg.pinv semantics for ill-conditioned Jacobians
// The reference solves the control law with numpy's pinv (experiment.py:312): SVD, singular values <= 1e-15 * sigma_max dropped.
// For full-rank J that equals the Householder least-squares solution the kernels compute; for (numerically) rank-deficient J it is
// the minimum-norm solution, which an unpivoted QR does not give.  Split of work:
//   * every least-squares solve watches two magnitudes of the triangular factor (all kernels; ~25 instructions per step in the headline
//     kernel): the smallest |R_cc| and the largest |R_ij| of the whole factor.  max |R_ij| <= sigma_max and min |R_cc| >= sigma_min, so their
//     ratio is a lower bound of the condition number; a ratio of 2^34 or more -- or an exactly zero column next to non-zero ones -- marks
//     the trial UVS_STATUS_SUSPECT.  That catches rank deficiency (a vanishing pivot) and bad column scaling, also where the two hide each
//     other: [[1, 1e20], [0, 1]] has an unremarkable diagonal and condition 1e40 (fixtures tests/golden/rankdef_gmckf_scaled_*: a column
//     scaled by 1e12 / 1e6 that is parallel to another within 1e-9 / 1e-12 -- numpy truncates, the diagonal alone shows a spread of 1e3).
//     What the spread alone does not see: a factor with every entry of ordinary size whose inverse still explodes (Kahan-like: unit diagonal,
//     all off-diagonals -1000, condition 3e18; fixture rankdef_gmckf_kahan).  Round 5 added the third watch for it -- the growth of the solution
//     against its right-hand side (Spread::grows below) -- and the fast kernels mark that trial too; UVS_OPT_STRICT_PINV (every trial through
//     the careful kernels) remains as a cross-check.
//     Normal-equation solvers (wide kernel, control wavefronts of the replay) mark at a spread of 2^20 already and are accurate to ~6e-9
//     in the command up to cond 1e6;
//   * suspect trials are re-run from their first step by the `careful` instantiation of the generic kernels, launched right behind
//     the main kernel by the C ABI, whose control law finishes the same QR with a one-sided Jacobi SVD of the n x n factor R
//     (pinv(J) y = pinv(R) Q^T y, same singular values as J) and applies numpy's cutoff.  The hot kernels stay free of that code.
constexpr int UVS_STATUS_SUSPECT = 2;            // internal: never visible after an entry point returns
constexpr unsigned kSuspectSpread = 68u << 20;   // exponent-field distance (high dword of a double) of R_cc^2: |R_cc| spread 2^34 ~ 1.7e10
constexpr double kPinvRcond = 1e-15;             // numpy.linalg.pinv default (rcond=1e-15)
constexpr unsigned kGrowthGate = (0x3ff00000u + kSuspectSpread) / 2;   // log2(max|sol|) + log2(max|R|) - log2(max|Q^T y|) >= 34, see Spread::grows

// Running exponent range of the non-negative doubles R_cc^2 (their high dwords order like the values; a zero column gives 0).
struct Spread {
    unsigned lo = 0xffffffffu, hi = 0u;
    UVS_DEV void add(double n2) {
        const unsigned e = (unsigned)__double2hiint(n2);
        lo = e < lo ? e : lo;
        hi = e > hi ? e : hi;
    }
    // The largest magnitude mx >= 0 among the entries of the factor (running fmax of |R_ij|; NaNs drop out of fmax and are caught by the
    // column norms): its square's high dword is 2 hi(mx) - 0x3ff00000 up to the mantissa product, i.e. within a factor 2 -- plenty for
    // a 2^34 gate.  Only raises `hi`: the smallest pivot stays the diagonal's.
    UVS_DEV void add_largest(double mx) {
        const unsigned h = (unsigned)__double2hiint(mx);
        unsigned e2 = h >= 0x20000000u ? 2u * h - 0x3ff00000u : 0u;             // (below 2^-511 the square underflows: no information)
        e2 = e2 > 0x7ff00000u ? 0x7ff00000u : e2;                               // a square beyond the range reads as +inf ("suspect"), never as NaN ("non-finite")
        hi = e2 > hi ? e2 : hi;
    }
    // Solution growth (round 5; call after add_largest): max |sol_c| * max |R_ij| / max |(Q^T y)_c| >= 2^34, in the exponent fields like everything
    // here (hi holds the SQUARE of the largest entry, hence the shift; within a factor 8).  sigma_max / sigma_min is at least this ratio, and
    // unlike the spread of the factor's entries it also sees a triangle of unremarkable entries whose inverse explodes (Kahan-like: unit
    // diagonal, off-diagonals -1000, condition 3e18) -- whenever the right-hand side excites the small direction, which is exactly when the
    // plain least-squares command and numpy's truncated one differ materially.  On the reference fixtures the ratio is <= 4e2 for every healthy
    // step (49 closed-loop / fixed-point fixtures) and >= 2.4e15 on every step of rankdef_gmckf_kahan_c1000 (tests/growth_watch_study.py).
    UVS_DEV bool grows(double smax, double cmax) const {
#ifdef UVS_NO_GROWTH_WATCH              // experiment builds: A/B of the watch's cost (its inputs then fold away)
        return false;
#endif
        const unsigned hs = (unsigned)__double2hiint(smax), hc = (unsigned)__double2hiint(cmax);
        return hs + (hi >> 1) >= hc + kGrowthGate;                              // (every term below 2^31: no wrap)
    }
    // lo == 0: a column vanished altogether (all of them when hi == 0 too, J = 0, where the plain solve would divide 0 by 0)
    UVS_DEV bool suspect() const { return hi - lo >= kSuspectSpread || lo == 0u; }
};

// sol = pinv(Rm) c for a small square matrix through a one-sided Jacobi (Hestenes) SVD: columns of A = Rm are rotated until
// mutually orthogonal, A = U diag(sigma), V accumulates the rotations, Rm = U diag(sigma) V^T.  numpy's cutoff: components with
// sigma_j <= 1e-15 * sigma_max are dropped.  Only the careful kernels instantiate this (rare path; clarity over speed).
template <int N>
UVS_DEV void svd_solve(double (&A)[N][N], const double (&c)[N], double (&sol)[N]) {
    double V[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < N - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    alpha = fma(A[i][p], A[i][p], alpha);
                    beta = fma(A[i][q], A[i][q], beta);
                    gamma = fma(A[i][p], A[i][q], gamma);
                }
                if (fabs(gamma) > 2.3e-16 * sqrt(alpha * beta) && gamma != 0.0) {
                    rotated = true;
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double t = (fabs(zeta) > 1e150) ? 0.5 / zeta : copysign(1.0, zeta) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
                    const double cs = 1.0 / sqrt(fma(t, t, 1.0)), sn = cs * t;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const double ap = A[i][p], aq = A[i][q];
                        A[i][p] = cs * ap - sn * aq;
                        A[i][q] = sn * ap + cs * aq;
                        const double vp = V[i][p], vq = V[i][q];
                        V[i][p] = cs * vp - sn * vq;
                        V[i][q] = sn * vp + cs * vq;
                    }
                }
            }
        }
        if (!__any(rotated)) break;
    }
    double s2[N], proj[N], s2max = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) { a = fma(A[i][j], A[i][j], a); b = fma(A[i][j], c[i], b); }
        s2[j] = a;
        proj[j] = b;                                              // sigma_j * (u_j . c)
        s2max = a > s2max ? a : s2max;
    }
    const double cut = kPinvRcond * sqrt(s2max);
#pragma unroll
    for (int i = 0; i < N; ++i) sol[i] = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const bool large = sqrt(s2[j]) > cut;                     // numpy: s > rcond * max(s)
        const double w = large ? proj[j] / s2[j] : 0.0;           // (u_j . c) / sigma_j
#pragma unroll
        for (int i = 0; i < N; ++i) sol[i] = fma(V[i][j], w, sol[i]);
    }
}

// ---------------------------------------------------------------- normal equations (wide-shape kernel, control-law wavefronts of the replay)
// G = J^T J = L L^T.  The pivots of the Cholesky factor are the R_cc^2 of the QR of J, so the same spread test marks ill-conditioned
// Jacobians for the careful second pass -- here already from a spread of 2^20 in |R_cc|, because the error of the normal equations
// grows with cond(J)^2 (one refinement step squares it again: measured against numpy's pinv 5e-14 relative at cond <= 1.5e3, the same
// as Householder, 3e-11 at cond 1e5, 6e-9 at cond 1e6).
constexpr unsigned kSuspectSpreadNormalEq = 40u << 20;
constexpr unsigned kRefineGate = 20u << 20;      // refinement correction >= 2^-20 of the solution (high dwords): the normal equations have run out of digits

// In place: G[at(j, i)], i > j, becomes L_ij; rs[j] = 1 / L_jj.  Returns the suspect verdict: pivots spread too far, or one of them
// is not a positive normal number (zero / negative: breakdown; inf / NaN).
// COLS = false leaves the column norms out of the watch (pivots only): the wide-shape kernel sits at 505 of 512 registers and spills with them.
template <int N, bool COLS = true>
UVS_DEV bool chol_factor(double (&G)[Sym<N>::NP], double (&rs)[N]) {
    Spread spread;
    // the largest squared column norm of J (the diagonal of G before it is touched) bounds sigma_max^2 from below just like max R_ij^2 does
    // in the QR solvers -- and costs no arithmetic and no live range here (tracking max |L_ij| cost the wide kernel 30 registers and spills)
    unsigned col_hi = 0u;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = G[Sym<N>::at(j, j)];
        if constexpr (COLS) {                                    // squared norm of column j of J, read where it is consumed anyway
            const unsigned e = (unsigned)__double2hiint(d);
            col_hi = (e > col_hi && e <= 0x7ff00000u) ? e : col_hi;   // (a NaN norm is left to the pivot, which turns NaN with it)
        }
#pragma unroll
        for (int k = 0; k < j; ++k) d = fma(-G[Sym<N>::at(k, j)], G[Sym<N>::at(k, j)], d);
        spread.add(d);
        double sq, r;
        fast_sqrt_rsqrt(d, sq, r);
        rs[j] = r;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = G[Sym<N>::at(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) v = fma(-G[Sym<N>::at(k, i)], G[Sym<N>::at(k, j)], v);
            G[Sym<N>::at(j, i)] = v * r;
        }
    }
    spread.hi = col_hi > spread.hi ? col_hi : spread.hi;
    return spread.hi - spread.lo >= kSuspectSpreadNormalEq || spread.lo == 0u || spread.hi >= 0x7ff00000u;
}

// L z = b, L^T x = z in place; L[at(j, i)] for i > j holds L_ij, rs[j] = 1 / L_jj
template <int N>
UVS_DEV void chol_solve_inplace(const double (&L)[Sym<N>::NP], const double (&rs)[N], double (&b)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double v = b[j];
#pragma unroll
        for (int k = 0; k < j; ++k) v = fma(-L[Sym<N>::at(k, j)], b[k], v);
        b[j] = v * rs[j];
    }
#pragma unroll
    for (int j = N - 1; j >= 0; --j) {
        double v = b[j];
#pragma unroll
        for (int i = j + 1; i < N; ++i) v = fma(-L[Sym<N>::at(j, i)], b[i], v);
        b[j] = v * rs[j];
    }
}

// ---------------------------------------------------------------- control law: dq = -gain * pinv(J) y
// Overdetermined / square case (M >= N): Householder QR of [J | y] distributed over the L lanes of the
// group (each lane holds R rows), then back substitution.  Equals numpy's pinv(J) @ y (experiment.py:312)
// for full-column-rank J; exactly-zero columns give a zero component (pinv(0) = 0).
template <int M, int N, int L, bool CAREFUL = false>
UVS_DEV bool lstsq_tall(double (&a)[M / L][N + 1], int sub, double (&sol)[N]) {
    constexpr int R = M / L;
    double diag[N];
    double rcc[CAREFUL ? N : 1];                                 // R_cc, kept only for the careful finish
    double rmax = 0.0;                                           // largest |R_ij| seen off the diagonal
    Spread spread;
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const int owner = c / R, prow = c % R;                    // lane / local row holding global row c
        double sig = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool below = (L == 1) ? (r > c) : (sub * R + r > c);
            sig = below ? fma(a[r][c], a[r][c], sig) : sig;
        }
        sig = group_sum<L>(sig);
        const double piv = group_pick<L>(a[prow][c], sub, owner);
        const double n2 = fma(piv, piv, sig);
        double nrm, rn;
        fast_sqrt_rsqrt_1(n2 > 0.0 ? n2 : 1.0, nrm, rn);          // |R_cc| and its reciprocal (a zero column must not breed NaNs)
        spread.add(n2);
        const double alpha = (n2 > 0.0) ? ((piv >= 0.0) ? -nrm : nrm) : 0.0;
        if constexpr (CAREFUL) rcc[c] = alpha;
        const double vp = piv - alpha;                            // pivot entry of the Householder vector
        const double denom = n2;                                  // zero column <=> nothing to eliminate
        const double tau = (denom > 0.0) ? rn * fast_rcp_1(fabs(vp)) : 0.0;   // 2 / (v.v) = 1 / (nrm (nrm + |piv|))
        const bool mine = (L == 1) ? true : (sub == owner);
#pragma unroll
        for (int j = c + 1; j <= N; ++j) {
            double d = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool below = (L == 1) ? (r > c) : (sub * R + r > c);
                d = below ? fma(a[r][c], a[r][j], d) : d;
            }
            d = mine ? fma(vp, a[prow][j], d) : d;
            d = group_sum<L>(d) * tau;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool below = (L == 1) ? (r > c) : (sub * R + r > c);
                a[r][j] = below ? fma(-d, a[r][c], a[r][j]) : a[r][j];
            }
            a[prow][j] = mine ? fma(-d, vp, a[prow][j]) : a[prow][j];
            if (j < N) rmax = fmax(rmax, fabs(a[prow][j]));          // row c of R on its owner (elsewhere: some row not larger than its column's norm)
        }
        diag[c] = (denom > 0.0) ? ((piv >= 0.0) ? -rn : rn) : 0.0;  // 1 / R_cc (0 marks a zero column)
    }
    spread.add_largest(group_max<L>(rmax));
    const bool suspect = spread.suspect();
    if constexpr (CAREFUL) {
        // pinv(J) y = pinv(R) (Q^T y)[0:N]: gather the triangular factor and the transformed right-hand side on every lane of the group
        // and finish with the SVD -- on EVERY solve of a careful kernel, not only when the |R_cc| watch fires again: the diagonal of an
        // unpivoted QR is not rank-revealing (J = [[1, 1e20], [0, 1]] has spread 1 and condition 1e40), so once a trial is in the careful
        // pass nothing is left to that heuristic.  For a well-conditioned factor the SVD solve equals the back substitution to rounding.
        {
            double Rm[N][N], cv[N], ss[N];
#pragma unroll
            for (int c = 0; c < N; ++c) {
                const int owner = c / R, prow = c % R;
#pragma unroll
                for (int j = 0; j < N; ++j) Rm[c][j] = (j < c) ? 0.0 : (j == c ? rcc[c] : group_pick<L>(a[prow][j], sub, owner));
                cv[c] = group_pick<L>(a[prow][N], sub, owner);
            }
            svd_solve<N>(Rm, cv, ss);
#pragma unroll
            for (int c = 0; c < N; ++c) sol[c] = ss[c];
            return suspect;
        }
    }
#pragma unroll
    for (int c = N - 1; c >= 0; --c) {
        const int owner = c / R, prow = c % R;
        double rhs = a[prow][N];
#pragma unroll
        for (int j = c + 1; j < N; ++j) rhs = fma(-a[prow][j], sol[j], rhs);
        rhs = group_pick<L>(rhs, sub, owner);
        sol[c] = rhs * diag[c];
    }
    double smax = 0.0, cmax = 0.0;                               // solution growth (Spread::grows): the top N entries of Q^T y against the solution
#pragma unroll
    for (int c = 0; c < N; ++c) smax = fmax(smax, fabs(sol[c]));
#pragma unroll
    for (int r = 0; r < R; ++r) cmax = fmax(cmax, (sub * R + r < N) ? fabs(a[r][N]) : 0.0);
    return suspect || spread.grows(smax, group_max<L>(cmax));
}

// Underdetermined case (M < N, e.g. one feature: 2 x 6): minimum-norm solution through the QR of J^T.
// Only instantiated with L == 1 (the whole J in one lane).
template <int M, int N, bool CAREFUL = false>
UVS_DEV bool lstsq_wide(const double (&J)[M][N], const double (&y)[M], double (&sol)[N]) {
    double b[N][M];                                               // J^T, overwritten by R (upper) and v (lower)
    double vp[M], tau[M], diag[M];
    double wmax = 0.0;
    Spread spread;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) b[j][i] = J[i][j];
#pragma unroll
    for (int c = 0; c < M; ++c) {
        double sig = 0.0;
#pragma unroll
        for (int r = c + 1; r < N; ++r) sig = fma(b[r][c], b[r][c], sig);
        const double piv = b[c][c];
        const double nrm = sqrt(fma(piv, piv, sig));
        const double alpha = (piv >= 0.0) ? -nrm : nrm;
        const double denom = nrm * (nrm + fabs(piv));
        vp[c] = piv - alpha;
        tau[c] = (denom > 0.0) ? 1.0 / denom : 0.0;
        diag[c] = (denom > 0.0) ? alpha : piv;
        spread.add(diag[c] * diag[c]);
#pragma unroll
        for (int j = c + 1; j < M; ++j) {
            double d = vp[c] * b[c][j];
#pragma unroll
            for (int r = c + 1; r < N; ++r) d = fma(b[r][c], b[r][j], d);
            d *= tau[c];
            b[c][j] = fma(-d, vp[c], b[c][j]);
            wmax = fmax(wmax, fabs(b[c][j]));
#pragma unroll
            for (int r = c + 1; r < N; ++r) b[r][j] = fma(-d, b[r][c], b[r][j]);
        }
    }
    spread.add_largest(wmax);
    // R^T w = y (forward substitution), then sol = Q [w; 0] = H_0 .. H_{M-1} [w; 0]
    double w[N];
#pragma unroll
    for (int j = 0; j < N; ++j) w[j] = 0.0;
    const bool suspect = spread.suspect();
    bool solved = false;
    if constexpr (CAREFUL) {
        // J = [R^T 0] Q^T, so pinv(J) y = Q [pinv(R^T) y; 0]: the m x m factor goes through the SVD with numpy's cutoff
        {                                                          // every solve of a careful kernel (see lstsq_tall)
            double Rt[M][M], ws[M];
#pragma unroll
            for (int c = 0; c < M; ++c)
#pragma unroll
                for (int j = 0; j < M; ++j) Rt[c][j] = (j > c) ? 0.0 : (j == c ? diag[c] : b[j][c]);
            svd_solve<M>(Rt, y, ws);
#pragma unroll
            for (int c = 0; c < M; ++c) w[c] = ws[c];
            solved = true;
        }
    }
    if (!solved) {
#pragma unroll
        for (int c = 0; c < M; ++c) {
            double rhs = y[c];
#pragma unroll
            for (int j = 0; j < c; ++j) rhs = fma(-b[j][c], w[j], rhs);
            w[c] = (diag[c] != 0.0) ? rhs / diag[c] : 0.0;
        }
    }
#pragma unroll
    for (int c = M - 1; c >= 0; --c) {
        double d = vp[c] * w[c];
#pragma unroll
        for (int r = c + 1; r < N; ++r) d = fma(b[r][c], w[r], d);
        d *= tau[c];
        w[c] = fma(-d, vp[c], w[c]);
#pragma unroll
        for (int r = c + 1; r < N; ++r) w[r] = fma(-d, b[r][c], w[r]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) sol[j] = w[j];
    return suspect;
}

// dq = -gain * pinv(X.reshape(m, n)) @ (kappa o err)   (experiment.py:300-312); result replicated in the group
// Returns true when the solve looked rank-deficient (see "numpy.linalg.pinv semantics" above); with CAREFUL the result is pinv's then.
template <int M, int N, int L, bool CAREFUL = false>
UVS_DEV bool control_law(const Rows<M, N, L> &st, const double (&kap)[M / L], const double (&err)[M / L], double gain, int sub,
                         double (&dq)[N]) {
    constexpr int R = M / L;
    double sol[N];
    bool suspect;
    if constexpr (M >= N) {
        double a[R][N + 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < N; ++j) a[r][j] = st.x[r][j];
            a[r][N] = kap[r] * err[r];
        }
        suspect = lstsq_tall<M, N, L, CAREFUL>(a, sub, sol);
    } else {
        static_assert(M >= N || L == 1, "wide Jacobians are handled one filter per lane");
        double y[R];
#pragma unroll
        for (int r = 0; r < R; ++r) y[r] = kap[r] * err[r];
        suspect = lstsq_wide<M, N, CAREFUL>(st.x, y, sol);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) dq[j] = -gain * sol[j];
    return suspect;
}

// ---------------------------------------------------------------- plant
// Camera pose T_0_N of the DH chain at joints q: rot (row-major 3x3) and pos (ur10_simulation.py:97-110, 204-211).
// When `zs`/`ps` are given they receive the z axis / origin of every intermediate frame for the Jacobian.
template <int N, bool WITH_FRAMES>
UVS_DEV void forward_kinematics(const uvs_plant &pl, const double (&q)[N], double (&rot)[9], double (&pos)[3],
                                double (*zs)[3], double (*ps)[3]) {
    double T[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s, c;
        sincos(q[i] + pl.theta_offset[i], &s, &c);
        const double ca = pl.cos_alpha[i], sa = pl.sin_alpha[i], aa = pl.a[i], dd = pl.d[i];
        // link = Rz(theta) Tz(d) Rx(alpha) Tx(a)
        const double l01 = -s * ca, l02 = s * sa, l03 = aa * c;
        const double l11 = c * ca, l12 = -c * sa, l13 = aa * s;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double t0 = T[r][0], t1 = T[r][1], t2 = T[r][2], t3 = T[r][3];
            T[r][0] = fma(t0, c, t1 * s);
            T[r][1] = fma(t0, l01, fma(t1, l11, t2 * sa));
            T[r][2] = fma(t0, l02, fma(t1, l12, t2 * ca));
            T[r][3] = fma(t0, l03, fma(t1, l13, fma(t2, dd, t3)));
        }
        if constexpr (WITH_FRAMES) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { zs[i][r] = T[r][2]; ps[i][r] = T[r][3]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) rot[3 * r + cidx] = T[r][cidx];
        pos[r] = T[r][3];
    }
}

// Pinhole image coordinate `axis` (0 = u, 1 = v) of world point w seen from (rot, pos); depth = |pos - w| on request.
UVS_DEV double project_axis(const double (&rot)[9], const double (&pos)[3], const double *w, int axis, double focal, double center) {
    const double dx = w[0] - pos[0], dy = w[1] - pos[1], dz = w[2] - pos[2];
    const double xc = fma(rot[0], dx, fma(rot[3], dy, rot[6] * dz));       // R^T (w - t)
    const double yc = fma(rot[1], dx, fma(rot[4], dy, rot[7] * dz));
    const double zc = fma(rot[2], dx, fma(rot[5], dy, rot[8] * dz));
    return center + focal * (axis == 0 ? xc : yc) / zc;
}

// Noise-free features of this lane's R rows at joints q.
template <int M, int N, int L>
UVS_DEV void plant_features(const uvs_plant &pl, const double (&q)[N], int sub, double (&f)[M / L]) {
    constexpr int R = M / L;
    if (pl.kind == UVS_PLANT_LINEAR) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = sub * R + r;
            double acc = pl.lin_f0[row];
#pragma unroll
            for (int j = 0; j < N; ++j) acc = fma(pl.lin_jacobian[row * N + j], q[j] - pl.lin_q0[j], acc);
            f[r] = acc;
        }
        return;
    }
    double rot[9], pos[3];
    forward_kinematics<N, false>(pl, q, rot, pos, nullptr, nullptr);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = sub * R + r;
        f[r] = project_axis(rot, pos, pl.points[row >> 1], row & 1, pl.focal, pl.center);
    }
}

// Analytic initial guess X0 = J_img kron(I2, R^T) J_robot for this lane's rows, plus the noise-free f
// at q (experiment.py:86-114; geometric Jacobian ur10_simulation.py:112-139).
template <int M, int N, int L>
UVS_DEV void initial_guess(const uvs_plant &pl, const double (&q)[N], int sub, double (&x)[M / L][N], double (&f)[M / L]) {
    constexpr int R = M / L;
    double rot[9], pos[3], zs[N][3], ps[N][3];
    forward_kinematics<N, true>(pl, q, rot, pos, zs, ps);
    double Jc[6][N];                                                        // camera-frame twist Jacobian kron(I2, R^T) J
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double z[3], o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            z[r] = (i == 0) ? (r == 2 ? 1.0 : 0.0) : zs[i - 1][r];
            o[r] = (i == 0) ? 0.0 : ps[i - 1][r];
        }
        const double dx = pos[0] - o[0], dy = pos[1] - o[1], dz = pos[2] - o[2];
        const double lin[3] = {z[1] * dz - z[2] * dy, z[2] * dx - z[0] * dz, z[0] * dy - z[1] * dx};   // z x (p_e - p)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Jc[r][i] = fma(rot[r], lin[0], fma(rot[3 + r], lin[1], rot[6 + r] * lin[2]));
            Jc[3 + r][i] = fma(rot[r], z[0], fma(rot[3 + r], z[1], rot[6 + r] * z[2]));
        }
    }
    const double F = pl.focal;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = sub * R + r;
        const double *w = pl.points[row >> 1];
        const double u = project_axis(rot, pos, w, 0, F, pl.center);
        const double v = project_axis(rot, pos, w, 1, F, pl.center);
        const double ddx = pos[0] - w[0], ddy = pos[1] - w[1], ddz = pos[2] - w[2];
        const double depth = sqrt(fma(ddx, ddx, fma(ddy, ddy, ddz * ddz)));  // computeZ: |cam - disc|
        double ji[6];
        if ((row & 1) == 0) {                                                // experiment.py:101-109 (u row)
            ji[0] = -F / depth; ji[1] = 0.0; ji[2] = u / depth; ji[3] = u * v / F; ji[4] = -(F * F + u * u) / F; ji[5] = v;
            f[r] = u;
        } else {                                                             // experiment.py:102-110 (v row)
            ji[0] = 0.0; ji[1] = -F / depth; ji[2] = v / depth; ji[3] = (F * F + v * v) / F; ji[4] = -u * v / F; ji[5] = -u;
            f[r] = v;
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc = fma(ji[k], Jc[k][j], acc);
            x[r][j] = acc;
        }
    }
}

}  // namespace uvs

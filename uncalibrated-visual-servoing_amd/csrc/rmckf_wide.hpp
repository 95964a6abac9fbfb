// Tuned closed-loop kernel for the wide stress shape (BASELINE config 5: m = 32 rows, n = 7 joints, linear consistent plant).
//
// Sizing.  A filter is 32 covariance blocks of 28 doubles + 224 doubles of X = 8.96 KB (the reference's dense 224 x 224 P would be
// 401 KB and cannot exist on chip at all).  L = 8 lanes share a filter (lanes 8 apart: lane = 8 sub + trial), lane `sub` owning rows sub, sub + 8, sub + 16, sub + 24:
// 140 doubles of state per lane, so one wavefront per SIMD (512 registers; what does not fit the 256 VALU-addressable ones the compiler
// parks in AGPRs), 8 trials per wavefront.
//
// What differs from the generic template that served this shape before (45-52 ms per 65 536 x 299 sweep):
//   * control law by the normal equations with one refinement step instead of Householder QR across 16 lanes.  Each lane accumulates
//     J^T J and J^T y over its own rows, ONE batch of 36 independent group sums follows (the QR needs 35 sums too, but one after the
//     other, each on the critical path, plus a replicated sqrt / reciprocal chain per column) -- at L = 8 as a reduce-scatter /
//     all-gather over v_permlane32_swap / v_permlane16_swap (blocked_sums8 below) --, then every lane factors the 7 x 7 Gram matrix
//     itself; the refinement step costs 7 more sums.  Accuracy: rmckf_device.hpp, "normal equations";
//   * estimator selected at compile time, rows through the same rmckf_row as the (8,6) kernels;
//   * plant matrix rows and desired features in LDS, joints replicated on the 8 lanes (no exchange), stream cursors instead of
//     per-element address arithmetic, next step's noise fetched one step ahead.
#pragma once
#include "rmckf_tuned.hpp"

namespace uvs {

// Scheduling fence between the rows of a lane: without it the compiler interleaves the four row updates for instruction-level
// parallelism, needs ~700 registers and spills to scratch; one wavefront per SIMD gains nothing from that interleaving anyway.
#define UVS_WIDE_FENCE() __builtin_amdgcn_sched_barrier(0)

// v[idx] for a per-lane idx on a register-resident array (a select chain; a variably indexed access would go through scratch memory)
template <int N>
UVS_DEV double pick_lane_value(const double (&v)[N], int idx) {
    double r = v[0];
#pragma unroll
    for (int j = 1; j < N; ++j) r = (idx == j) ? in_reg(v[j]) : r;
    return r;
}

// ---- sums over the 8 lanes of a filter in the BLOCKED lane mapping (L == 8: lane = 8 sub + trial, so the lanes of a filter sit 8 / 16 / 32 apart).
// The control law needs 36 sums per step on every lane of the filter.  As 36 butterfly all-reduces (the interleaved mapping: three DPP stages of
// 2 moves + 1 add each) that is 324 instructions; here it is a reduce-scatter / all-gather: v_permlane32_swap exchanges the upper half of one
// register with the lower half of another, so ONE pair of swaps + one add folds lane bit 5 of TWO values at once (each half of the wavefront keeps
// one of them), v_permlane16_swap does the same for lane bit 4 between the 16-lane rows, the last stage (lane bit 3) is a plain DPP row rotate by 8
// on the quarter of the values that is left; two gather stages (copy + swap) hand the totals back.  18 x 3 + 9 x 3 + 9 x 3 + 9 x 4 + 18 x 4 = 216.
// Every lane of a filter receives the same bits (each total is computed once, then copied).
UVS_DEV void swap_halves(double &a, double &b) {                   // a = [a.lo | b.lo], b = [a.hi | b.hi]   (halves of 32 lanes)
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
UVS_DEV void swap_rows(double &a, double &b) {                     // a = [a.r0, b.r0, a.r2, b.r2], b = [a.r1, b.r1, a.r3, b.r3]   (rows of 16 lanes)
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
constexpr int kDppRowRor8 = 0x128;
template <int NV>
UVS_DEV void blocked_sums8(double (&v)[NV]) {
    constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2;
    double u[N1], w[N2];
#pragma unroll
    for (int i = 0; i < N1; ++i) {                                 // lane bit 5: lower half keeps v[2 i], upper half v[2 i + 1]
        double a = v[2 * i], b = v[2 * i + 1 < NV ? 2 * i + 1 : 2 * i];
        swap_halves(a, b);
        u[i] = a + b;
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) {                                 // lane bit 4: even rows keep u[2 i], odd rows u[2 i + 1]
        double a = u[2 * i], b = u[2 * i + 1 < N1 ? 2 * i + 1 : 2 * i];
        swap_rows(a, b);
        w[i] = a + b;
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) w[i] += dpp_mov64<kDppRowRor8>(w[i]);   // lane bit 3
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        double a = w[i], b = w[i];
        swap_rows(a, b);
        u[2 * i] = a;
        if (2 * i + 1 < N1) u[2 * i + 1] = b;
    }
#pragma unroll
    for (int i = 0; i < N1; ++i) {
        double a = u[i], b = u[i];
        swap_halves(a, b);
        v[2 * i] = a;
        if (2 * i + 1 < NV) v[2 * i + 1] = b;
    }
}
UVS_DEV double blocked_sum8(double v) {                            // one value: all-reduce over lane bits 5, 4, 3
    double a = v, b = v;
    swap_halves(a, b);
    v = a + b;
    a = v; b = v;
    swap_rows(a, b);
    v = a + b;
    return v + dpp_mov64<kDppRowRor8>(v);
}
template <int L>
UVS_DEV double wide_sum(double v) {
    if constexpr (L == 8) return blocked_sum8(v);
    else return group_sum<L>(v);
}

// XREC: the X stream is laid out as per-trial records ([step][trial][component]: comp_stride 1, trial_stride M N).  The TPW trials of a
// wavefront then own one contiguous block of TPW M N doubles per step (14 KB at (32,7), L = 8): the rows go through LDS once and leave as
// 16-byte-per-lane stores of 1 KB each, fully coalesced.  In the trial-fastest layout of the (8,6) kernels a wavefront of 8 trials can only
// write 64-byte pieces (8 trials x 8 B per component row), which is what held the generic template at 50 ms.
//
// PLANT = UVS_PLANT_DH_PINHOLE (round 4): the same kernel as an EIGHT-lane mapping of the (8,6) shape -- one row (one image coordinate) per lane, 8
// trials per wavefront, so that a shard of 8 192 trials (a rank's share of north_star's 65 536-trial series on 8 GPUs) is one round of 1 024
// wavefronts.  The kinematic chain is replicated on the 8 lanes of a filter (no exchange): joint sines / cosines carried from step to step by
// the addition theorems as in the tuned kernels, the chain applied right to left to the three columns a lane needs (its camera axis, the
// optical axis, the position).  Built as the latency mapping VERDICT r3 asked to be measured; measured: 8 192 trials 0.92 ms against 0.88 ms on
// four lanes per filter -- 1 075 VALU instructions per wavefront-step against 1 109 (PMC): the replicated plant (~300) gives back what one row per
// lane saves.  Reachable with lanes_per_filter = 8 (instead of the generic template); no launch policy selects it.
template <int M, int N, int L, int METHOD, bool XOUT, bool XREC, int PLANT = UVS_PLANT_LINEAR>
__global__ __launch_bounds__(64, 1) void closed_loop_wide_kernel(const ClosedArgs A) {
    static_assert(M % L == 0 && M >= N && (L == 8 || L == 16), "wide kernel: rows interleaved over 8 or 16 adjacent lanes (one DPP row)");
    constexpr int R = M / L, NP = Sym<N>::NP, TPW = 64 / L;
    constexpr bool DH = PLANT != UVS_PLANT_LINEAR;
    static_assert(!DH || (R == 1 && !XREC), "DH plant: one row per lane (initial_guess numbers a lane's rows sub R + r), streams in the caller's strides");
    constexpr int REC = M * N, RECP = REC + 1;                     // record length, padded in LDS against bank conflicts
    static_assert(!XREC || (TPW * REC) % 128 == 0, "record path: a whole number of 1 KB stores per wavefront");
    __shared__ double lJ[DH ? 1 : M][N];                           // plant matrix (linear plant)
    __shared__ double lc[DH ? 1 : M], ldes[M];                     // f0 - J q0, desired_f
    __shared__ double lt[XREC ? TPW * RECP : 1];                   // transposition buffer of the X records
    __shared__ double lacc[3 * R][64], lfp[R][64];                 // lane-private: ISE / IAE / ITAE accumulators, previous noisy features
    // KF and IMCC-KF weigh every row of a filter alike, so all their covariance blocks stay identical (RowShare, rmckf_tuned.hpp): a lane keeps
    // ONE block, its other rows only move their x.  RMCKF: PV of the lane's R blocks stay in registers, the others live in LDS and pass
    // through registers while their row is updated: with all four in registers (L = 8) the compiler overflows VGPRs + AGPRs and spills
    // ~60 dwords per lane to scratch.
    constexpr bool SHARED_P = (METHOD == UVS_METHOD_KF || METHOD == UVS_METHOD_IMCCKF);
    constexpr int PV = SHARED_P ? 1 : ((L == 8 && R > 1) ? R - 1 : R), PL = SHARED_P ? 0 : R - PV;
    __shared__ double lp[PL > 0 ? PL * NP : 1][64];

    const unsigned lane = threadIdx.x;
    // L == 8: blocked mapping, lane = TPW sub + trial (the sums above); L == 16: the lanes of a filter are one DPP row
    const int sub = (L == 8) ? (int)(lane / TPW) : (int)(lane & (L - 1));
    const long long wave_first = (long long)blockIdx.x * TPW;
    const unsigned tl = (L == 8) ? lane % TPW : lane / L;
    const bool valid = wave_first + tl < A.T;
    const long long trial = valid ? wave_first + tl : A.T - 1;     // padding lanes shadow the last trial (duplicate values, same addresses)
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;

    for (int i = (int)lane; i < M; i += 64) {
        if constexpr (!DH) {
            double c = A.plant.lin_f0[i];
            for (int j = 0; j < N; ++j) {
                const double v = A.plant.lin_jacobian[i * N + j];
                lJ[i][j] = v;
                c = fma(-v, A.plant.lin_q0[j], c);
            }
            lc[i] = c;
        }
        ldes[i] = fp.desired[i];
    }
    __syncthreads();

    // stream cursors of this lane's first row (components advance by L rows)
    const double *pn = A.noise.p ? A.noise.at(trial, 0, sub) : nullptr;
    // record path: every lane points at the first record of the wavefront (the last, ragged wavefront of a batch takes the plain path)
    double *px = (XOUT && A.x_out.p) ? (XREC ? A.x_out.at(wave_first, 0, 0) : A.x_out.at(trial, 0, sub * N)) : nullptr;
    double *pe = A.err_out.p ? A.err_out.at(trial, 0, sub) : nullptr;
    double *pf = A.f_out.p ? A.f_out.at(trial, 0, sub) : nullptr;
    double *pq = (A.q_out.p && sub < N) ? A.q_out.at(trial, 0, sub) : nullptr;           // lane `sub` logs joint `sub`
    double *pd = (A.dq_out.p && sub < N) ? A.dq_out.at(trial, 0, sub) : nullptr;

    double q[N], dq[N], x[R][N], p[PV][NP];
#pragma unroll
    for (int j = 0; j < N; ++j) { q[j] = *A.q_start.at(trial, 0, j); dq[j] = 0.0; }     // first_run: H = 0 (experiment.py:183-185)
    double f_first[R];
    bool guessed = false;
    if constexpr (DH) {
        if (fp.initial_guess) {                                    // X0 = analytic Jacobian at q_start, f = the noise-free features there (experiment.py:86-114)
            initial_guess<M, N, L>(A.plant, q, sub, x, f_first);
            guessed = true;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = r * L + sub;
        lfp[r][lane] = guessed ? f_first[r] : 0.0;                 // else f = zeros(m) (experiment.py:56)
        if (!guessed) {
#pragma unroll
            for (int j = 0; j < N; ++j) x[r][j] = *A.x0.at(trial, 0, row * N + j);
        }
#pragma unroll
        for (int l = 0; l < N; ++l)
#pragma unroll
            for (int j = l; j < N; ++j) {
                const double v = (l == j) ? 1.0 : 0.0;             // P = I (experiment.py:73)
                if (r < PV) p[r < PV ? r : 0][Sym<N>::at(l, j)] = v;
                else if constexpr (!SHARED_P) lp[(r >= PV ? r - PV : 0) * NP + Sym<N>::at(l, j)][lane] = v;
            }
        lacc[r][lane] = lacc[R + r][lane] = lacc[2 * R + r][lane] = 0.0;
    }
    // record path: where in the (padded) LDS copy the pair of doubles this lane stores with the i-th 1 KB store lives -- computed once,
    // the division by the record length does not belong in the step loop
    int src_off[XREC ? TPW * REC / 128 : 1];
    if constexpr (XREC) {
#pragma unroll
        for (int i = 0; i < TPW * REC / 128; ++i) {
            const int d = 128 * i + 2 * (int)lane;               // index inside the wavefront's block; a pair never straddles two records (REC is even)
            src_off[i] = (d / REC) * RECP + d % REC;
        }
    }
    double nz_next[R];
#pragma unroll
    for (int r = 0; r < R; ++r) nz_next[r] = (pn && K > 0) ? pn[(long long)r * L * A.noise.sc] : 0.0;
    if (pn) pn += A.noise.sk;
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): keep "nz_next may be in flight" out of the loop header

    double t = fp.dt;
    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true, flagged = false;
    // DH plant: the lane's image coordinate (row = sub: point sub / 2, u for even rows, v for odd ones) and the tracked sines / cosines
    double sn[DH ? N : 1], cs[DH ? N : 1], wpt[3] = {0.0, 0.0, 0.0};
    bool reseed = true;
    const bool odd = (sub & 1) != 0;
    if constexpr (DH) {
#pragma unroll
        for (int pt = 0; pt < M / 2; ++pt)
#pragma unroll
            for (int c = 0; c < 3; ++c) wpt[c] = ((sub >> 1) == pt) ? A.plant.points[pt][c] : wpt[c];
#pragma unroll
        for (int u = 0; u < N; ++u) { sn[u] = 0.0; cs[u] = 1.0; }
    }

    DiagPhases diag_steps;                                         // -DUVS_STAMPS build: per-phase cycle sums of every wavefront (tools/read_stamps.py --wide); `stats` is garbage then
    unsigned long long rt_first = 0;
    if constexpr (kDiagStamps) { diag_steps.last = diag_cycles(); rt_first = diag_ticks(); }
    for (int k = 0; k < K; ++k) {
        UVS_STAMP(7);                                              // loop edge: joint integration, clock, (DH: sincos advance)
        double nz[R];
#pragma unroll
        for (int r = 0; r < R; ++r) nz[r] = nz_next[r];
        if (pn && k + 1 < K) {                                     // next step's noise: a whole step to arrive
#pragma unroll
            for (int r = 0; r < R; ++r) nz_next[r] = pn[(long long)r * L * A.noise.sc];
            pn += A.noise.sk;
        }
        // ---- plant + measurement (experiment.py:134-135, 170-177, 302)
        double zi[R], err[R], kap[R];
        double z_dh = 0.0;
        if constexpr (DH) {
            // sines / cosines of the joint angles: carried by sincos_advance, re-seeded from the angle every kSinCosResync steps or at once after
            // a step too large for the polynomials (decided per lane; the lanes of a filter hold the same joints, so they decide alike)
            const bool need = reseed || (k & (kSinCosResync - 1)) == 0;
            if (__any(need)) {
                double th[N], s_new[N], c_new[N];
                bool big = false;
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    th[u] = q[u] + A.plant.theta_offset[u];
                    big |= !(fabs(th[u]) <= kSinCosBoundedMax);
                    sincos_bounded(th[u], s_new[u], c_new[u]);
                }
                if (__builtin_expect(__any(need && big), 0)) {
#pragma unroll
                    for (int u = 0; u < N; ++u) {
                        double sl, cl;
                        sincos(th[u], &sl, &cl);
                        s_new[u] = big ? sl : s_new[u];
                        c_new[u] = big ? cl : c_new[u];
                    }
                }
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    sn[u] = need ? s_new[u] : sn[u];
                    cs[u] = need ? c_new[u] : cs[u];
                }
                reseed = false;
            }
            // link = Rz(theta) Tz(d) Rx(alpha) Tx(a) (ur10_simulation.py:204-211); T = A_0 ... A_{N-1} applied right to left to three columns
            double va[3], vz[3], vp[3];
            {
                const double s = sn[N - 1], c = cs[N - 1], ca = A.plant.cos_alpha[N - 1], sa = A.plant.sin_alpha[N - 1], aa = A.plant.a[N - 1];
                va[0] = odd ? -s * ca : c; va[1] = odd ? c * ca : s; va[2] = odd ? sa : 0.0;
                vz[0] = s * sa; vz[1] = -c * sa; vz[2] = ca;
                vp[0] = aa * c; vp[1] = aa * s; vp[2] = A.plant.d[N - 1];
            }
#pragma unroll
            for (int i = N - 2; i >= 0; --i) {
                const double s = sn[i], c = cs[i], ca = A.plant.cos_alpha[i], sa = A.plant.sin_alpha[i], aa = A.plant.a[i], dd = A.plant.d[i];
                const double l01 = -s * ca, l02 = s * sa, l03 = aa * c, l11 = c * ca, l12 = -c * sa, l13 = aa * s;
                const double a0 = fma(c, va[0], fma(l01, va[1], l02 * va[2])), a1 = fma(s, va[0], fma(l11, va[1], l12 * va[2])), a2 = fma(sa, va[1], ca * va[2]);
                const double z0 = fma(c, vz[0], fma(l01, vz[1], l02 * vz[2])), z1 = fma(s, vz[0], fma(l11, vz[1], l12 * vz[2])), z2 = fma(sa, vz[1], ca * vz[2]);
                const double p0 = fma(c, vp[0], fma(l01, vp[1], fma(l02, vp[2], l03))), p1 = fma(s, vp[0], fma(l11, vp[1], fma(l12, vp[2], l13))),
                             p2 = fma(sa, vp[1], fma(ca, vp[2], dd));
                va[0] = a0; va[1] = a1; va[2] = a2; vz[0] = z0; vz[1] = z1; vz[2] = z2; vp[0] = p0; vp[1] = p1; vp[2] = p2;
            }
            const double dx = wpt[0] - vp[0], dy = wpt[1] - vp[1], dz = wpt[2] - vp[2];      // pinhole image coordinate: R^T (w - t), ur10_simulation.py:141-154
            const double ic = fma(va[0], dx, fma(va[1], dy, va[2] * dz));
            const double iz = fast_rcp(fma(vz[0], dx, fma(vz[1], dy, vz[2] * dz)));
            z_dh = fma(A.plant.focal * ic, iz, A.plant.center);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = r * L + sub;
            double f;
            if constexpr (DH) f = z_dh;
            else {
                f = lc[row];
#pragma unroll
                for (int j = 0; j < N; ++j) f = fma(lJ[row][j], q[j], f);
            }
            f += nz[r];
            zi[r] = f - lfp[r][lane];
            err[r] = f - ldes[row];
            lfp[r][lane] = f;
        }
        UVS_STAMP(0);                                              // noise issue + plant + measurement
        const double sigma = bandwidth(fp, k);
        const double neg_half_inv_s2 = -0.5 * fast_rcp(sigma * sigma);
        double c_shared = 1.0;
        if constexpr (METHOD == UVS_METHOD_IMCCKF) {               // one weight for the whole filter (experiment.py:258-261)
            double ss = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double pred = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) pred = fma(x[r][j], dq[j], pred);
                const double nu = zi[r] - pred;
                ss = fma(nu, nu, ss);
            }
            c_shared = exp_nonpos(wide_sum<L>(ss) * neg_half_inv_s2);
        }
        // ---- estimator rows (experiment.py:166-297)
        double chk = 0.0;
        double *pxr = px;
        RowShare<N> share;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            UVS_WIDE_FENCE();
            if constexpr (SHARED_P) {
                if (r == 0) {
                    NoHook none;
                    FpiProbe unused;
                    rmckf_row<N, METHOD>(x[0], p[0], dq, zi[0], neg_half_inv_s2, c_shared, fp.reg, kap[0], chk, unused, none, share);
                } else {
                    rmckf_row_follow<N>(x[r], share, dq, zi[r], chk);
                    kap[r] = 1.0;
                }
            } else if (r < PV) {
                rmckf_row<N, METHOD>(x[r], p[r < PV ? r : 0], dq, zi[r], neg_half_inv_s2, c_shared, fp.reg, kap[r], chk);
            } else {
                double pb[NP];
#pragma unroll
                for (int e = 0; e < NP; ++e) pb[e] = lp[(r >= PV ? r - PV : 0) * NP + e][lane];
                rmckf_row<N, METHOD>(x[r], pb, dq, zi[r], neg_half_inv_s2, c_shared, fp.reg, kap[r], chk);
#pragma unroll
                for (int e = 0; e < NP; ++e) lp[(r >= PV ? r - PV : 0) * NP + e][lane] = pb[e];
            }
            if constexpr (XOUT && !XREC) {
                if (pxr) {
                    double *pc = pxr;
#pragma unroll
                    for (int j = 0; j < N; ++j) { *pc = x[r][j]; pc += A.x_out.sc; }
                    pxr += (long long)L * N * A.x_out.sc;
                }
            }
            if constexpr (XOUT && XREC) {
#pragma unroll
                for (int j = 0; j < N; ++j) lt[tl * RECP + (r * L + sub) * N + j] = x[r][j];
            }
        }
        UVS_STAMP(1);                                              // row updates (+ X into the record buffer)
        UVS_WIDE_FENCE();
        // The wavefront's TPW records are one contiguous block: lane l stores doubles 2 l, 2 l + 1 of every 128-double (1 KB) slice.  Round 5: the
        // 14 stores no longer leave in one burst behind the rows (the stamped build showed the wavefront waiting 1 600 cycles per step at the full
        // store queue, and the next step's noise load -- vmcnt counts in order -- behind them): they are handed out in pieces to the control law
        // below, whose arithmetic they do not touch (the record buffer is rewritten only by the next step's rows).
        constexpr int NREC = XREC ? TPW * REC / 128 : 0;
        double *blk = (XOUT && XREC && px) ? px + 2 * lane : nullptr;
        auto store_records = [&](auto first, auto count) {
            if constexpr (XOUT && XREC) {
                if (blk) {
                    UVS_WIDE_FENCE();
#pragma unroll
                    for (int i = decltype(first)::value; i < decltype(first)::value + decltype(count)::value && i < NREC; ++i) {
                        const double *src = &lt[src_off[XREC ? i : 0]];
                        double2 v;
                        v.x = src[0];
                        v.y = src[1];
                        *reinterpret_cast<double2 *>(blk + 128 * i) = v;
                    }
                    UVS_WIDE_FENCE();
                }
            }
        };
        // the NREC stores in R + 3 pieces: one behind every Gram row, one behind the 36 sums, one behind the first solve, one in front of the refinement's
        // sums -- piece i covers records [i NREC / (R + 3), (i + 1) NREC / (R + 3)) (L = 8: 14 records in 7 pieces of 2; L = 16: 7 records in 5 pieces)
        auto piece = [&](auto idx) {
            constexpr int I = decltype(idx)::value, PIECES = R + 3;
            store_records(std::integral_constant<int, I * NREC / PIECES>{}, std::integral_constant<int, (I + 1) * NREC / PIECES - I * NREC / PIECES>{});
        };
        // Measured A/B on one box (profiles/r05/wide_phase_table.txt): KF 11.18 -> 10.82 ms with the pieces, RMCKF 14.07 -> 14.23 with them -- its lone,
        // VALU-bound wavefront loses more to the fences that pin the pieces than the queue gives back -- so RMCKF keeps the burst.
        constexpr bool SPREAD_STORES = SHARED_P;
        if constexpr (!SPREAD_STORES) store_records(std::integral_constant<int, 0>{}, std::integral_constant<int, NREC>{});
        (void)pxr;
        if (px) px += A.x_out.sk;
        UVS_STAMP(2);                                              // record stores
        // ---- control law dq = -gain pinv(X)(kappa o err) (experiment.py:300-312): normal equations + one refinement step
        double G[NP], b[N], y[R];
#pragma unroll
        for (int e = 0; e < NP; ++e) G[e] = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) b[j] = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            UVS_WIDE_FENCE();
            y[r] = kap[r] * err[r];
#pragma unroll
            for (int l = 0; l < N; ++l) {
#pragma unroll
                for (int j = l; j < N; ++j) G[Sym<N>::at(l, j)] = fma(x[r][l], x[r][j], G[Sym<N>::at(l, j)]);
                b[l] = fma(x[r][l], y[r], b[l]);
            }
            if constexpr (SPREAD_STORES) {                         // piece r of R + 3 (see piece() above)
                if (r == 0) piece(std::integral_constant<int, 0>{});
                if (r == 1) piece(std::integral_constant<int, 1>{});
                if (r == 2) piece(std::integral_constant<int, 2>{});
                if (r == 3) piece(std::integral_constant<int, 3>{});
                static_assert(R <= 4, "one piece per Gram row");
            }
        }
        if constexpr (L == 8) {                                    // one batch: Gram matrix, right-hand side and the finiteness probe
            double v[NP + N + 1];
#pragma unroll
            for (int e = 0; e < NP; ++e) v[e] = G[e];
#pragma unroll
            for (int j = 0; j < N; ++j) v[NP + j] = b[j];
            v[NP + N] = chk;
            blocked_sums8(v);
#pragma unroll
            for (int e = 0; e < NP; ++e) G[e] = v[e];
#pragma unroll
            for (int j = 0; j < N; ++j) b[j] = v[NP + j];
            chk = v[NP + N];
        } else {
            chk = group_sum<L>(chk);
#pragma unroll
            for (int e = 0; e < NP; ++e) G[e] = group_sum<L>(G[e]);    // independent sums: they pipeline
#pragma unroll
            for (int j = 0; j < N; ++j) b[j] = group_sum<L>(b[j]);
        }
        UVS_STAMP(3);                                              // Gram matrix + right-hand side + their 36 sums over the filter's lanes
        if constexpr (SPREAD_STORES) piece(std::integral_constant<int, R>{});
        if (alive && !(chk == 0.0)) {                              // X turned non-finite: pinv would raise (experiment.py:313-316)
            alive = false;
            status = UVS_STATUS_FAIL;
            k_done = k;
        }
        if (!__any(alive)) break;
        UVS_WIDE_FENCE();
        double rs[N];
        const bool suspect = chol_factor<N, DH>(G, rs);          // (wide shape: pivot spread only -- no register left for the column-norm watch, see chol_factor; the (8,6) latency kernel has them)
        chol_solve_inplace<N>(G, rs, b);                           // s0
        if constexpr (SPREAD_STORES) piece(std::integral_constant<int, R + 1>{});
        double c[N];
#pragma unroll
        for (int j = 0; j < N; ++j) c[j] = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {                              // c = J^T (y - J s0)
            double ri = y[r];
#pragma unroll
            for (int j = 0; j < N; ++j) ri = fma(-x[r][j], b[j], ri);
#pragma unroll
            for (int j = 0; j < N; ++j) c[j] = fma(x[r][j], ri, c[j]);
        }
        if constexpr (SPREAD_STORES) piece(std::integral_constant<int, R + 2>{});
        if constexpr (L == 8) blocked_sums8(c);
        else {
#pragma unroll
            for (int j = 0; j < N; ++j) c[j] = group_sum<L>(c[j]);
        }
        chol_solve_inplace<N>(G, rs, c);
        // Refinement watch (round 6; the normal equations' answer to Spread::grows of the QR solvers).  A Kahan-like Jacobian (unit-diagonal
        // triangle, large off-diagonals: condition 1e19) leaves every Cholesky PIVOT at an ordinary value -- the spread test above sees nothing, and
        // because squaring J has already destroyed its smallest singular value, the solution does not even grow much.  What gives it away is that
        // the refinement step does not converge: the correction is cond(J)^2 eps of the solution -- at most 7e-14 on healthy (32,7) closed loops,
        // at least 2e-3 on every step of Kahan-like ones that does not break the factorisation outright (tests/growth_watch_study.py --wide).
        // Gate: |correction| >= 2^-20 |s0|, i.e. cond(J) ~ 1e5, a little ahead of the pivot gate (2^20); numpy's pinv (experiment.py:312)
        // truncates from 1e15, and the careful pass that redoes a marked trial decides that by SVD.
        {
            double s_max = 0.0, c_max = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) { s_max = fmax(s_max, fabs(b[j])); c_max = fmax(c_max, fabs(c[j])); }
            const bool stalls = (unsigned)__double2hiint(c_max) + kRefineGate >= (unsigned)__double2hiint(s_max) && c_max > 0.0;
            flagged |= alive && (suspect || stalls);               // ill-conditioned Jacobian: the careful second pass redoes this trial
        }
#pragma unroll
        for (int j = 0; j < N; ++j) dq[j] = -fp.gain * (b[j] + c[j]);

        UVS_STAMP(5);                                              // Cholesky + solve + refinement (7 more sums) + second solve
        // ---- logs and statistics
        if (pe) {
            double *pc = pe;
#pragma unroll
            for (int r = 0; r < R; ++r) { *pc = err[r]; pc += (long long)L * A.err_out.sc; }
            pe += A.err_out.sk;
        }
        if (pf) {
            double *pc = pf;
#pragma unroll
            for (int r = 0; r < R; ++r) { *pc = lfp[r][lane]; pc += (long long)L * A.f_out.sc; }
            pf += A.f_out.sk;
        }
        if (pq) { *pq = pick_lane_value<N>(q, sub); pq += A.q_out.sk; }
        if (pd) { *pd = pick_lane_value<N>(dq, sub); pd += A.dq_out.sk; }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double e = alive ? err[r] : 0.0;                 // a failed trial stops contributing
            const double ae = fabs(e);
            lacc[r][lane] = fma(e, e, lacc[r][lane]);
            lacc[R + r][lane] += ae;
            lacc[2 * R + r][lane] = fma(t, ae, lacc[2 * R + r][lane]);
        }
        UVS_STAMP(6);                                              // logs + statistics
        if constexpr (DH) {
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const double d = dq[u] * fp.dt;
                reseed |= !(fabs(d) <= kSinCosStepMax);            // too large for the polynomials, or not finite: re-seed at the next step
                sincos_advance(sn[u], cs[u], d);
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) q[j] = fma(dq[j], fp.dt, q[j]);            // new_q = q + dq t_s (experiment.py:320)
        t += fp.dt;
    }

    if constexpr (kDiagStamps) {
        diag_steps.sum[4] = diag_ticks() - rt_first;
        if (lane == 0 && A.stats) {
            for (int c = 0; c < 8; ++c) A.stats[3 * wave_first + c] = (double)diag_steps.sum[c];
        }
        return;
    }
    double s2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) v = fma(lacc[c * R + r][lane], lacc[c * R + r][lane], v);
        s2[c] = wide_sum<L>(v);
    }
    if (!valid) return;
    if (sub == 0) {
        if (A.stats) {
#pragma unroll
            for (int c = 0; c < 3; ++c) A.stats[3 * trial + c] = sqrt(s2[c]);
        }
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
                for (int j = 0; j < N; ++j)
                    *A.p_final.at(trial, 0, ((r * L + sub) * N + l) * N + j) =
                        SHARED_P ? p[0][Sym<N>::at(l, j)]
                                 : (r < PV) ? p[r < PV ? r : 0][Sym<N>::at(l, j)] : lp[(r >= PV ? r - PV : 0) * NP + Sym<N>::at(l, j)][lane];
    }
}

}  // namespace uvs

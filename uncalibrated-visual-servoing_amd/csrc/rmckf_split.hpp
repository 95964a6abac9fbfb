// Role-split closed-loop kernel (lanes_per_filter = 5; an EXPERIMENT that is measured and kept as a variant, not the default).
// A 256-thread workgroup (4 wavefronts, one per SIMD) owns 64 trials and runs every step in two phases.
//
//   estimator phase   all 4 wavefronts, 4 lanes per filter (lane = sub * 16 + trial, rows sub and sub + 4): the row updates of
//                     experiment.py:166-297.  Rows of a filter never talk to each other, the covariance blocks stay in registers, X and
//                     kappa o err go to LDS.
//   control turn      ONE wavefront, one lane per trial (64 trials): dq = -gain pinv(X)(kappa o err) (experiment.py:300-312) by
//                     normal equations + one refinement step streamed from LDS, joint update, kinematic chain + pinhole projection of
//                     the next step (ur10_simulation.py:97-110).  Lane-local: no cross-lane reductions, no duplicated scalar chains,
//                     64 trials per instruction instead of the 32 of the two-lane kernel.  Meanwhile the other three wavefronts stream
//                     X / err of the finished step from LDS to HBM (512 contiguous bytes per store) and fetch the next noise.
//                     The role rotates (step k: wavefront (k + block) mod 4) so that the serial work lands evenly on the 4 SIMDs.
//
// Motive: the two-lane kernel (rmckf_tuned.hpp) spends 59 % of its ~1545 VALU instructions per wavefront-step outside the filter rows
// (plant and QR replicate work across the lanes of a filter or pay a DPP reduction per dot product) and runs one wavefront per SIMD.
// Per trial this kernel needs ~37 wavefront-instructions per step instead of 48.
//
// Measured (MI355X, config 2, `make split_stamps` + tools/read_split_stamps.py): 4.98 ms against 3.29 ms for the two-lane kernel.  Of the
// 18.4 k cycles of a step the control turn takes 14.8 k (normal equations 5.4 k, q / dq logs 2.4 k, kinematics 6.4 k): ~1400 instructions
// at ~10 cycles each, because the 256-register budget of two workgroups per CU forces rolled loops with exposed LDS latency (unrolled,
// the compiler spills 100-200 registers to scratch), the turn is serial by construction, and only two workgroups per CU (80 KB of LDS,
// 256 registers) are there to overlap it: SIMD issue utilisation ends at ~34 %.  What would be needed to win -- a control turn of ~5 k
// cycles -- is ~4 cycles per instruction on a SIMD shared with another wavefront; see DESIGN.md section 4.
//
// LDS per workgroup (doubles, [component][trial]; strides of the arrays the estimator role touches are padded so that its 2 x 16-lane
// halves hit disjoint banks): lx 48 x 72, lrhs / lerr / lzf / lnz 8 x 80, ldq / lq 6 x 64, lchk 4 x 64, lacc 6 x 256, lpark 21 x 64:
// 80 032 B, two workgroups per CU.
#pragma once
#ifndef UVS_SPLIT_OCC
#define UVS_SPLIT_OCC 2
#endif
#include "rmckf_tuned.hpp"

namespace uvs {

// One trial per lane: dq-direction sol = pinv(J) y for the M x N Jacobian held in LDS as lx[(i N + j)][lane] and the right-hand side
// lrhs[i][lane], by the normal equations with one step of iterative refinement:
//     G = J^T J = L L^T (Cholesky);  s0 = G^-1 J^T y;  sol = s0 + G^-1 J^T (y - J s0).
// Why not the Householder QR of the other kernels: lane-local it needs the whole 8 x 7 panel in registers (112 + ~60) next to the
// wavefront's covariance blocks (84) -- over the 256 a wavefront may hold with two workgroups per CU.  This form streams J from LDS
// twice and keeps 27 + 21 doubles.  One refinement step squares the cond(J)^2 eps error of the plain normal equations: measured against
// numpy's pinv on the reference's own trajectories (cond <= 1.5e3) 5e-14 relative, the same as Householder (2e-14 ... 7e-14), 3e-11 at
// cond 1e5, 6e-9 at cond 1e6.  The pivots of the Cholesky factor are the R_cc^2 of the QR, so the same spread test marks
// ill-conditioned Jacobians for the careful second pass, here already from a spread of 2^20 in |R_cc|.
template <int M, int N, int SX, int SV>
UVS_DEV bool lstsq_normal_lds(const double (*lx)[SX], const double (*lrhs)[SV], unsigned lane, double (&sol)[N]) {
    constexpr int NP = Sym<N>::NP;
    double G[NP], b[N];
#pragma unroll
    for (int e = 0; e < NP; ++e) G[e] = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) b[j] = 0.0;
    // One row of J per iteration of a ROLLED loop, the next row's LDS reads in flight under this row's arithmetic.  Fully unrolled, the
    // compiler hoists all 56 reads to the top, interleaves the rows and spills the wavefront's covariance blocks to scratch to make room
    // (measured: ~210 live registers in this function instead of ~90).
    const double *px = &lx[0][lane], *py = &lrhs[0][lane];
    double xn[N + 1];
#pragma unroll
    for (int j = 0; j < N; ++j) xn[j] = px[j * SX];
    xn[N] = py[0];
#pragma unroll 1
    for (int i = 0; i < M; ++i) {                                 // G += x_i x_i^T, b += x_i y_i
        double xi[N + 1];
#pragma unroll
        for (int j = 0; j <= N; ++j) xi[j] = xn[j];
        if (i + 1 < M) {
            px += N * SX;
            py += SV;
#pragma unroll
            for (int j = 0; j < N; ++j) xn[j] = px[j * SX];
            xn[N] = py[0];
        }
#pragma unroll
        for (int l = 0; l < N; ++l) {
#pragma unroll
            for (int j = l; j < N; ++j) G[Sym<N>::at(l, j)] = fma(xi[l], xi[j], G[Sym<N>::at(l, j)]);
            b[l] = fma(xi[l], xi[N], b[l]);
        }
    }
    double rs[N];
    const bool suspect = chol_factor<N>(G, rs);
    chol_solve_inplace<N>(G, rs, b);                             // s0
    double c[N];
#pragma unroll
    for (int j = 0; j < N; ++j) c[j] = 0.0;
    px = &lx[0][lane];
    py = &lrhs[0][lane];
#pragma unroll
    for (int j = 0; j < N; ++j) xn[j] = px[j * SX];
    xn[N] = py[0];
#pragma unroll 1
    for (int i = 0; i < M; ++i) {                                 // c = J^T (y - J s0)
        double xi[N + 1];
#pragma unroll
        for (int j = 0; j <= N; ++j) xi[j] = xn[j];
        if (i + 1 < M) {
            px += N * SX;
            py += SV;
#pragma unroll
            for (int j = 0; j < N; ++j) xn[j] = px[j * SX];
            xn[N] = py[0];
        }
        double ri = xi[N];
#pragma unroll
        for (int j = 0; j < N; ++j) ri = fma(-xi[j], b[j], ri);
#pragma unroll
        for (int j = 0; j < N; ++j) c[j] = fma(xi[j], ri, c[j]);
    }
    chol_solve_inplace<N>(G, rs, c);
#pragma unroll
    for (int j = 0; j < N; ++j) sol[j] = b[j] + c[j];
    return suspect;
}

// Noise-free features of all M rows of one trial per lane at the joints lq[.][lane]: DH chain (ur10_simulation.py:97-110, 204-211) and
// pinhole projection, plant constants from LDS (PlantLds layout).  The joint loop is ROLLED on purpose: unrolled, the compiler evaluates
// the six sincos side by side (~120 registers of temporaries) and spills the wavefront's covariance blocks to scratch.
template <int M, int N, int SQ>
UVS_DEV void plant_dh_local(const double *lds_c, const double (*lq)[SQ], unsigned lane, double (&z)[M]) {
    using PC = PlantLds<M, N>;
    bool big = false;
#pragma unroll
    for (int u = 0; u < N; ++u) big |= !(fabs(lq[u][lane] + lds_c[PC::kJoint + 5 * u]) <= kSinCosBoundedMax);
    const bool any_big = __any(big);                            // one range test per step: bounded sincos or the library routine
    double T[3][4] = {{1.0, 0.0, 0.0, 0.0}, {0.0, 1.0, 0.0, 0.0}, {0.0, 0.0, 1.0, 0.0}};
    const double *cj = &lds_c[PC::kJoint];
    const double *pq = &lq[0][lane];
#pragma unroll 1
    for (int u = 0; u < N; ++u) {
        const double th = pq[0] + cj[0];
        double s, c;
        if (__builtin_expect(any_big, 0)) sincos(th, &s, &c);
        else sincos_bounded(th, s, c);
        const double dd = cj[1], aa = cj[2], ca = cj[3], sa = cj[4];
        const double l01 = -s * ca, l02 = s * sa, l03 = aa * c;    // link = Rz(theta) Tz(d) Rx(alpha) Tx(a)
        const double l11 = c * ca, l12 = -c * sa, l13 = aa * s;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double t0 = T[r][0], t1 = T[r][1], t2 = T[r][2], t3 = T[r][3];
            T[r][0] = fma(t0, c, t1 * s);
            T[r][1] = fma(t0, l01, fma(t1, l11, t2 * sa));
            T[r][2] = fma(t0, l02, fma(t1, l12, t2 * ca));
            T[r][3] = fma(t0, l03, fma(t1, l13, fma(t2, dd, t3)));
        }
        cj += 5;
        pq += SQ;
    }
    const double focal = lds_c[PC::kCam], center = lds_c[PC::kCam + 1];
#pragma unroll
    for (int pt = 0; pt < M / 2; ++pt) {
        const double dx = lds_c[PC::kPoint + 3 * pt] - T[0][3], dy = lds_c[PC::kPoint + 3 * pt + 1] - T[1][3],
                     dz = lds_c[PC::kPoint + 3 * pt + 2] - T[2][3];
        const double xc = fma(T[0][0], dx, fma(T[1][0], dy, T[2][0] * dz));           // R^T (w - t)
        const double yc = fma(T[0][1], dx, fma(T[1][1], dy, T[2][1] * dz));
        const double iz = fast_rcp(fma(T[0][2], dx, fma(T[1][2], dy, T[2][2] * dz)));
        z[2 * pt] = fma(focal * xc, iz, center);
        z[2 * pt + 1] = fma(focal * yc, iz, center);
    }
}

// Address of element (trial, step k, component c) of a stream: the (k, c) part is wavefront-uniform (scalar registers), the trial part a
// 32-bit per-lane byte offset computed once per stream -- global_load/store "saddr + voffset" form, no per-stream cursor registers are
// carried through the step.  The launcher only picks this kernel when (T - 1) * trial_stride * 8 fits 32 bits for every stream.
UVS_DEV double *stream_at(const View &v, unsigned trial_bytes, int k, int c) {
    char *row = reinterpret_cast<char *>(v.p + ((long long)k * v.sk + (long long)c * v.sc));
    return reinterpret_cast<double *>(row + trial_bytes);
}
UVS_DEV unsigned trial_offset(const View &v, long long trial) { return (unsigned)(trial * v.st * 8); }

// Diagnostic build -DUVS_SPLIT_STAMPS: per-wavefront cycle sums (s_memtime) of the phases of a step, written over the workgroup's slice of
// `stats` (which is garbage in that build): [wave][slot], slots 0 estimator phase, 1 wait at barrier 1, 2 control turn, 3 logging turn,
// 4 books, 5 wait at barrier 2, 6 loop wall ticks (s_memrealtime, 100 MHz), 7 total cycles.  Never part of the shipped library.
#ifdef UVS_SPLIT_STAMPS
#define UVS_SSTAMP(slot)                                                                  \
    do {                                                                                  \
        unsigned long long now_;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");      \
        stamp_sum[slot] += now_ - stamp_last;                                             \
        stamp_last = now_;                                                                \
    } while (0)
#else
#define UVS_SSTAMP(slot) do { } while (0)
#endif

template <int M, int N, int METHOD, bool XOUT>
__global__ __launch_bounds__(256, UVS_SPLIT_OCC) void closed_loop_split_kernel(const ClosedArgs A) {
    static_assert(M % 4 == 0 && M >= N, "split kernel: 4 estimator lanes per filter, tall Jacobian");
    constexpr int LE = 4, RE = M / LE, NP = Sym<N>::NP, TPB = 64, TPW = 16;
    constexpr int SX = 72, SV = 80;                               // padded component strides (see header)
    using PC = PlantLds<M, N>;
    __shared__ double lx[M * N][SX];                              // X_k
    __shared__ double lrhs[M][SV], lerr[M][SV];                   // kappa_k o err_k (control law), err_k (logs, statistics)
    __shared__ double lzf[M][SV], lnz[M][SV];                     // noise-free features and measurement noise of the coming step
    __shared__ double ldq[N][TPB], lq[N][TPB], lchk[LE][TPB];     // command, joints, finiteness probes
    __shared__ double lacc[3 * RE][4 * 64];                       // ISE / IAE / ITAE accumulators of every estimator lane
    __shared__ double lpark[NP][TPB];                             // one covariance block of the control wavefront's lanes, parked for its turn
    __shared__ int lflag[TPB];
    __shared__ double lds_c[PC::kCount + M];                      // plant constants, then desired_f
    constexpr int kDes = PC::kCount;

    const unsigned tid = threadIdx.x, lane = tid & 63;
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));      // wavefront index, in a scalar register
    const int sub = (int)(lane >> 4);                             // estimator role: row group
    const unsigned et = w * TPW + (lane & 15);                    // estimator role: trial within the workgroup
    const long long first = (long long)blockIdx.x * TPB;
    const bool valid_e = first + et < A.T;
    const long long trial_e = valid_e ? first + et : A.T - 1;     // padding lanes shadow the last trial (duplicate values, same addresses)
    const long long trial_c = (first + lane < A.T) ? first + lane : A.T - 1;      // control / logging role: one trial per lane
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;
    const unsigned rot = blockIdx.x & 3u;

    if (tid < (unsigned)N) {
        lds_c[PC::kJoint + 5 * tid + 0] = A.plant.theta_offset[tid];
        lds_c[PC::kJoint + 5 * tid + 1] = A.plant.d[tid];
        lds_c[PC::kJoint + 5 * tid + 2] = A.plant.a[tid];
        lds_c[PC::kJoint + 5 * tid + 3] = A.plant.cos_alpha[tid];
        lds_c[PC::kJoint + 5 * tid + 4] = A.plant.sin_alpha[tid];
    }
    if (tid < (unsigned)(M / 2)) {
#pragma unroll
        for (int c = 0; c < 3; ++c) lds_c[PC::kPoint + 3 * tid + c] = A.plant.points[tid][c];
    }
    if (tid == 0) { lds_c[PC::kCam] = A.plant.focal; lds_c[PC::kCam + 1] = A.plant.center; }
    if (tid < (unsigned)M) lds_c[kDes + tid] = fp.desired[tid];
    if (tid < (unsigned)TPB) lflag[tid] = 0;
#pragma unroll
    for (int i = 0; i < 3 * RE; ++i) lacc[i][tid] = 0.0;
    __syncthreads();

    const bool on_noise = A.noise.p != nullptr, on_x = XOUT && A.x_out.p != nullptr, on_err = A.err_out.p != nullptr,
               on_q = A.q_out.p != nullptr, on_f = A.f_out.p != nullptr, on_dq = A.dq_out.p != nullptr;
    const unsigned o_nz = trial_offset(A.noise, trial_c), o_x = trial_offset(A.x_out, trial_c), o_e = trial_offset(A.err_out, trial_c),
                   o_q = trial_offset(A.q_out, trial_c), o_d = trial_offset(A.dq_out, trial_c);

    // ---- prologue by the wavefront that "ran the control phase of step -1": initial state, features and noise of step 0
    if (w == ((rot + 3u) & 3u)) {
        double q[N], f0[M], z[M];
#pragma unroll
        for (int j = 0; j < N; ++j) q[j] = *A.q_start.at(trial_c, 0, j);
        if (fp.initial_guess) {
            double xa[M][N];
            initial_guess<M, N, 1>(A.plant, q, 0, xa, f0);         // analytic X0 and the noise-free f before the loop (experiment.py:86-114)
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) lx[i * N + j][lane] = xa[i][j];
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                f0[i] = 0.0;                                       // f = zeros(m) (experiment.py:56)
#pragma unroll
                for (int j = 0; j < N; ++j) lx[i * N + j][lane] = *A.x0.at(trial_c, 0, i * N + j);
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) { lq[j][lane] = q[j]; ldq[j][lane] = 0.0; }        // first_run: H = 0 (experiment.py:183-185)
        plant_dh_local<M, N, TPB>(lds_c, lq, lane, z);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            lzf[i][lane] = z[i];
            lnz[i][lane] = (on_noise && K > 0) ? *stream_at(A.noise, o_nz, 0, i) : 0.0;
            lerr[i][lane] = f0[i];                                 // handed to the estimator lanes as f_old of the first step
        }
    }
    __syncthreads();

    // ---- estimator state of this lane that stays in registers: covariance blocks and last noisy features of rows sub, sub + 4 of
    // trial et (X lives in lx)
    double p[RE][NP], f_prev[RE];
#pragma unroll
    for (int r = 0; r < RE; ++r) {
        f_prev[r] = lerr[r * LE + sub][et];
#pragma unroll
        for (int l = 0; l < N; ++l)
#pragma unroll
            for (int j = l; j < N; ++j) p[r][Sym<N>::at(l, j)] = (l == j) ? 1.0 : 0.0;     // P = I (experiment.py:73)
    }
    double t = fp.dt;
    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true;
    __syncthreads();                                               // lerr is rewritten by the first estimator phase

#ifdef UVS_SPLIT_STAMPS
    unsigned long long stamp_sum[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last, stamp_first, rt_first;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_first)::"memory");
    stamp_first = stamp_last;
#endif
    for (int k = 0; k < K; ++k) {
        const unsigned cw = (rot + (unsigned)k) & 3u;              // control wavefront of this step
        const bool control = (w == cw);

        // ================================================================ estimator phase (all wavefronts)
        {
            double dq[N], x[RE][N], zi[RE], err[RE], kap[RE];
#pragma unroll
            for (int j = 0; j < N; ++j) dq[j] = ldq[j][et];
#pragma unroll
            for (int r = 0; r < RE; ++r) {
                const int row = r * LE + sub;
                const double fi = lzf[row][et] + lnz[row][et];    // noisy feature (experiment.py:134-135)
                zi[r] = fi - f_prev[r];                            // measurement Z (experiment.py:170-177)
                err[r] = fi - lds_c[kDes + row];                   // experiment.py:302
                f_prev[r] = fi;
                if (on_f) *stream_at(A.f_out, trial_offset(A.f_out, trial_e), k, row) = fi;
#pragma unroll
                for (int j = 0; j < N; ++j) x[r][j] = lx[row * N + j][et];
            }
            const double sigma = bandwidth(fp, k);
            const double neg_half_inv_s2 = -0.5 * fast_rcp(sigma * sigma);
            double c_shared = 1.0;
            if constexpr (METHOD == UVS_METHOD_IMCCKF) {           // one weight for the whole filter (experiment.py:258-261)
                double ss = 0.0;
#pragma unroll
                for (int r = 0; r < RE; ++r) {
                    double pred = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) pred = fma(x[r][j], dq[j], pred);
                    const double nu = zi[r] - pred;
                    ss = fma(nu, nu, ss);
                }
                ss += __shfl_xor(ss, 16, 64);                      // lanes lane ^ 16, lane ^ 32 hold the other row groups of the trial
                ss += __shfl_xor(ss, 32, 64);
                c_shared = exp_nonpos(ss * neg_half_inv_s2);
            }
            double chk = 0.0;
#pragma unroll
            for (int r = 0; r < RE; ++r) {
                rmckf_row<N, METHOD>(x[r], p[r], dq, zi[r], neg_half_inv_s2, c_shared, fp.reg, kap[r], chk);
                const int row = r * LE + sub;
#pragma unroll
                for (int j = 0; j < N; ++j) lx[row * N + j][et] = x[r][j];
                lrhs[row][et] = kap[r] * err[r];
                lerr[row][et] = err[r];
            }
            lchk[sub][et] = chk;
        }
        UVS_SSTAMP(0);
        __syncthreads();                                           // X_k, kappa_k o err_k, chk_k visible to the control / logging roles
        UVS_SSTAMP(1);

        // ================================================================ control phase (one wavefront) | logging (the others)
        if (control) {
            // The control role needs ~150 registers next to the ~100 of estimator state this wavefront keeps: its second covariance
            // block waits in LDS meanwhile (the compiler's own answer is 48 dwords of scratch traffic per lane and turn).
#pragma unroll
            for (int e = 0; e < NP; ++e) lpark[e][lane] = p[RE - 1][e];
            UVS_SSTAMP(8);
            // ---- dq_k = -gain pinv(X_k)(kappa_k o err_k), lane = trial (experiment.py:300-312)
            const bool ok = ((lchk[0][lane] + lchk[1][lane]) + (lchk[2][lane] + lchk[3][lane])) == 0.0;
            double sol[N];
            const bool suspect = lstsq_normal_lds<M, N, SX, SV>(lx, lrhs, lane, sol);
            if (ok && suspect) lflag[lane] = 1;                    // ill-conditioned Jacobian: the careful second pass redoes this trial
            UVS_SSTAMP(9);
            double q[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double cmd = -fp.gain * sol[j];
                q[j] = lq[j][lane];
                if (on_q) *stream_at(A.q_out, o_q, k, j) = q[j];                // q_log[k] (experiment.py:323)
                if (on_dq) *stream_at(A.dq_out, o_d, k, j) = cmd;
                ldq[j][lane] = cmd;                                // regressor of step k + 1 (experiment.py:188)
                q[j] = fma(cmd, fp.dt, q[j]);                      // new_q = q + dq t_s (experiment.py:320)
                lq[j][lane] = q[j];
            }
            UVS_SSTAMP(10);
            if (k + 1 < K) {                                       // noise-free features of step k + 1
                double z[M];
                plant_dh_local<M, N, TPB>(lds_c, lq, lane, z);
#pragma unroll
                for (int i = 0; i < M; ++i) lzf[i][lane] = z[i];
            }
            UVS_SSTAMP(11);
#pragma unroll
            for (int e = 0; e < NP; ++e) p[RE - 1][e] = lpark[e][lane];
            UVS_SSTAMP(12);
        } else {
            // ---- logs of step k from LDS, one trial per lane: 512 contiguous bytes per store; the three logging wavefronts share the
            // M N + M component rows round-robin.  The last of them also fetches the noise of step k + 1 (requested first: vmcnt counts
            // in order, so the wait in front of the LDS writes leaves the younger stores in flight)
            const unsigned rank = (w - cw - 1u) & 3u;              // 0, 1, 2 among the non-control wavefronts
            double nz[M];
            const bool fetch = rank == 2u && on_noise && k + 1 < K;
            if (fetch) {
#pragma unroll
                for (int i = 0; i < M; ++i) nz[i] = *stream_at(A.noise, o_nz, k + 1, i);
            }
            if (on_x) {
#pragma unroll
                for (int i = 0; i < (M * N + 2) / 3; ++i) {
                    const unsigned c = 3u * i + rank;
                    if (c < (unsigned)(M * N)) *stream_at(A.x_out, o_x, k, (int)c) = lx[c][lane];
                }
            }
            if (on_err) {
#pragma unroll
                for (int i = 0; i < (M + 2) / 3; ++i) {
                    const unsigned c = 3u * i + rank;
                    if (c < (unsigned)M) *stream_at(A.err_out, o_e, k, (int)c) = lerr[c][lane];
                }
            }
            if (fetch) {
#pragma unroll
                for (int i = 0; i < M; ++i) lnz[i][lane] = nz[i];
            }
            UVS_SSTAMP(3);
        }
        // ---- every wavefront closes the books of its own estimator rows: finiteness verdict, statistics (off the critical path for
        // three of the four; the control wavefront does it after its serial work)
        {
            const double chk_t = (lchk[0][et] + lchk[1][et]) + (lchk[2][et] + lchk[3][et]);
            if (alive && !(chk_t == 0.0)) {                        // X turned non-finite: pinv would raise (experiment.py:313-316)
                alive = false;
                status = UVS_STATUS_FAIL;
                k_done = k;
            }
#pragma unroll
            for (int r = 0; r < RE; ++r) {
                const double e = alive ? lerr[r * LE + sub][et] : 0.0;            // a failed trial stops contributing
                const double ae = fabs(e);
                lacc[r][tid] = fma(e, e, lacc[r][tid]);
                lacc[RE + r][tid] += ae;
                lacc[2 * RE + r][tid] = fma(t, ae, lacc[2 * RE + r][tid]);
            }
        }
        t += fp.dt;
        UVS_SSTAMP(4);
        __syncthreads();                                           // z_{k+1}, noise_{k+1}, dq_k visible; lx / lrhs / lerr / lchk free again
        UVS_SSTAMP(5);
    }
#ifdef UVS_SPLIT_STAMPS
    {
        unsigned long long rt_last, now_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_last)::"memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");
        stamp_sum[6] = rt_last - rt_first;
        stamp_sum[7] = now_ - stamp_first;
        if (lane == 0 && A.stats && first + 64 <= A.T) {
            for (int c = 0; c < 8; ++c) A.stats[3 * first + w * 8 + c] = (double)stamp_sum[c];
            for (int c = 8; c < 13; ++c) A.stats[3 * first + 40 + w * 5 + (c - 8)] = (double)stamp_sum[c];
            unsigned id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
            A.stats[3 * first + 32 + w] = (double)id;
        }
        return;
    }
#endif

    // ---- per-trial results: the 4 lanes of a trial combine their rows' statistics through LDS (lchk is free now)
    double s2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < RE; ++r) v = fma(lacc[c * RE + r][tid], lacc[c * RE + r][tid], v);
        s2[c] = v;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lchk[sub][et] = s2[c];
        __syncthreads();
        s2[c] = (lchk[0][et] + lchk[1][et]) + (lchk[2][et] + lchk[3][et]);
        __syncthreads();
    }
    if (!valid_e) return;
    if (sub == 0) {
        if (A.stats) {
#pragma unroll
            for (int c = 0; c < 3; ++c) A.stats[3 * trial_e + c] = sqrt(s2[c]);
        }
        if (A.status) A.status[trial_e] = lflag[et] ? UVS_STATUS_SUSPECT : status;
        if (A.k_done) A.k_done[trial_e] = k_done;
    }
    if (A.x_final.on()) {
#pragma unroll
        for (int r = 0; r < RE; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) *A.x_final.at(trial_e, 0, (r * LE + sub) * N + j) = lx[(r * LE + sub) * N + j][et];
    }
    if (A.p_final.on()) {
#pragma unroll
        for (int r = 0; r < RE; ++r)
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = 0; j < N; ++j) *A.p_final.at(trial_e, 0, ((r * LE + sub) * N + l) * N + j) = p[r][Sym<N>::at(l, j)];
    }
}

}  // namespace uvs

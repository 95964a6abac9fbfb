// libuvs_rmckf.so -- kernels and C ABI (include/uvs_rmckf.h) of the batched RMCKF estimator, gfx950 only.
//
// Kernels (all fp64, wave64, 64-thread workgroups so that each wavefront is scheduled independently):
//   closed_loop_kernel<M,N,L>  whole servo trial per filter: plant -> noise -> estimator -> control law -> logs,
//                              K sequential steps with all state in registers (experiment.py:125-343 per trial,
//                              main.py:121-148 across trials).  Streams: noise in, X/err/q/f/dq out.
//   replay_kernel<M,N,L>       estimator + control law over recorded f / dq streams (experiment.py:166-312).
//   step_kernel<M,N,L>         one step with state in HBM (drop-in for a live robot behind Experiment.run()).
//   stats_kernel               ISE / IAE / ITAE norms of an error trajectory (results/plot_errorbar.m:39-84).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "rmckf_device.hpp"
#include "rmckf_tuned.hpp"
#include "rmckf_replay_tuned.hpp"
#include "noise_kernels.hpp"

namespace uvs {

template <int M, int N, int L>
UVS_DEV void store_final(const Rows<M, N, L> &st, const View &xf, const View &pf, long long trial, int sub) {
    constexpr int R = M / L;
    if (xf.on()) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) *xf.at(trial, 0, (sub * R + r) * N + j) = st.x[r][j];
    }
    if (pf.on()) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = 0; j < N; ++j) *pf.at(trial, 0, ((sub * R + r) * N + l) * N + j) = st.p[r][Sym<N>::at(l, j)];
    }
}

// ------------------------------------------------------------------------------------------------ closed loop
template <int M, int N, int L, int METHOD_T>
__global__ __launch_bounds__(64) void closed_loop_kernel(const ClosedArgs A) {
    constexpr int R = M / L;
    const long long gl = (long long)blockIdx.x * 64 + threadIdx.x;
    long long trial = gl / L;
    const int sub = (int)(gl % L);
    const bool valid = trial < A.T;
    if (!valid) trial = A.T - 1;                    // padding lanes shadow the last trial so group shuffles stay uniform
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;

    double q[N], dq[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { q[j] = *A.q_start.at(trial, 0, j); dq[j] = 0.0; }

    Rows<M, N, L> st;
    st.init_cov();
    double f_prev[R], des[R];
#pragma unroll
    for (int r = 0; r < R; ++r) des[r] = fp.desired[sub * R + r];
    if (fp.initial_guess) {
        initial_guess<M, N, L>(A.plant, q, sub, st.x, f_prev);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f_prev[r] = 0.0;                                                    // f = zeros(m) (experiment.py:56)
#pragma unroll
            for (int j = 0; j < N; ++j) st.x[r][j] = *A.x0.at(trial, 0, (sub * R + r) * N + j);
        }
    }

    double ise[R], iae[R], itae[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ise[r] = iae[r] = itae[r] = 0.0;
    double t = fp.dt;                               // start() steps the clock once (ur10_simulation.py:57)
    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true;

    double nz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) nz[r] = (A.noise.on() && K > 0) ? *A.noise.at(trial, 0, sub * R + r) : 0.0;

    for (int k = 0; k < K; ++k) {
        double nz_next[R];                          // prefetch the next step's noise under this step's arithmetic
#pragma unroll
        for (int r = 0; r < R; ++r) nz_next[r] = (A.noise.on() && k + 1 < K) ? *A.noise.at(trial, k + 1, sub * R + r) : 0.0;

        double f[R], z[R], err[R], kap[R];
        plant_features<M, N, L>(A.plant, q, sub, f);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f[r] += nz[r];                                                      // experiment.py:134-135
            z[r] = f[r] - f_prev[r];                                            // experiment.py:170-177
            f_prev[r] = f[r];
            err[r] = f[r] - des[r];                                             // experiment.py:302
        }
        st.template update<METHOD_T>(fp, z, dq, bandwidth(fp, k), kap);                            // h = previous command; zero on k = 0
        if (alive && st.any_nonfinite()) {                                      // pinv raises -> FAIL, break (experiment.py:313-316)
            alive = false;
            status = UVS_STATUS_FAIL;
            k_done = k;
        }
        if (!__any(alive)) break;
        control_law<M, N, L>(st, kap, err, fp.gain, sub, dq);

        if (alive && valid) {
            if (A.x_out.on()) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < N; ++j) *A.x_out.at(trial, k, (sub * R + r) * N + j) = st.x[r][j];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (A.err_out.on()) *A.err_out.at(trial, k, sub * R + r) = err[r];
                if (A.f_out.on()) *A.f_out.at(trial, k, sub * R + r) = f[r];
                const double ae = fabs(err[r]);
                ise[r] = fma(err[r], err[r], ise[r]);
                iae[r] += ae;
                itae[r] = fma(t, ae, itae[r]);
            }
            if (sub == 0) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (A.q_out.on()) *A.q_out.at(trial, k, j) = q[j];
                    if (A.dq_out.on()) *A.dq_out.at(trial, k, j) = dq[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) q[j] = fma(dq[j], fp.dt, q[j]);            // new_q = q + dq * t_s (experiment.py:320)
        t += fp.dt;
#pragma unroll
        for (int r = 0; r < R; ++r) nz[r] = nz_next[r];
    }

    double s2[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s2[0] = fma(ise[r], ise[r], s2[0]);
        s2[1] = fma(iae[r], iae[r], s2[1]);
        s2[2] = fma(itae[r], itae[r], s2[2]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) s2[i] = sqrt(group_sum<L>(s2[i]));
    if (valid) {
        store_final<M, N, L>(st, A.x_final, A.p_final, trial, sub);
        if (sub == 0) {
            if (A.stats) { A.stats[3 * trial] = s2[0]; A.stats[3 * trial + 1] = s2[1]; A.stats[3 * trial + 2] = s2[2]; }
            if (A.status) A.status[trial] = status;
            if (A.k_done) A.k_done[trial] = k_done;
        }
    }
}

// ------------------------------------------------------------------------------------------------ replay
template <int M, int N, int L, int METHOD_T>
__global__ __launch_bounds__(64) void replay_kernel(const ReplayArgs A) {
    constexpr int R = M / L;
    const long long gl = (long long)blockIdx.x * 64 + threadIdx.x;
    long long trial = gl / L;
    const int sub = (int)(gl % L);
    const bool valid = trial < A.T;
    if (!valid) trial = A.T - 1;
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;

    Rows<M, N, L> st;
    st.init_cov();
    double f_prev[R], des[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        des[r] = fp.desired[sub * R + r];
        f_prev[r] = *A.f.at(trial, 0, sub * R + r);
#pragma unroll
        for (int j = 0; j < N; ++j) st.x[r][j] = *A.x0.at(trial, 0, (sub * R + r) * N + j);
    }
    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true;
    for (int k = 0; k < K; ++k) {
        double f[R], z[R], err[R], kap[R], h[N], cmd[N];
#pragma unroll
        for (int j = 0; j < N; ++j) h[j] = (k == 0) ? 0.0 : *A.dq.at(trial, k, j);   // first_run: H = 0 (experiment.py:183)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f[r] = *A.f.at(trial, k + 1, sub * R + r);
            z[r] = f[r] - f_prev[r];
            f_prev[r] = f[r];
            err[r] = f[r] - des[r];
        }
        st.template update<METHOD_T>(fp, z, h, bandwidth(fp, k), kap);
        if (alive && st.any_nonfinite()) {
            alive = false;
            status = UVS_STATUS_FAIL;
            k_done = k;
        }
        if (!__any(alive)) break;
        control_law<M, N, L>(st, kap, err, fp.gain, sub, cmd);
        if (alive && valid) {
            if (A.x_out.on()) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < N; ++j) *A.x_out.at(trial, k, (sub * R + r) * N + j) = st.x[r][j];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (A.err_out.on()) *A.err_out.at(trial, k, sub * R + r) = err[r];
                if (A.kappa_out.on()) *A.kappa_out.at(trial, k, sub * R + r) = kap[r];
            }
            if (sub == 0 && A.dqcmd_out.on()) {
#pragma unroll
                for (int j = 0; j < N; ++j) *A.dqcmd_out.at(trial, k, j) = cmd[j];
            }
        }
    }
    if (valid) {
        store_final<M, N, L>(st, A.x_final, A.p_final, trial, sub);
        if (sub == 0) {
            if (A.status) A.status[trial] = status;
            if (A.k_done) A.k_done[trial] = k_done;
        }
    }
}

// ------------------------------------------------------------------------------------------------ single step
template <int M, int N, int L, int METHOD_T>
__global__ __launch_bounds__(64) void step_kernel(const StepArgs A) {
    constexpr int R = M / L;
    const long long gl = (long long)blockIdx.x * 64 + threadIdx.x;
    long long trial = gl / L;
    const int sub = (int)(gl % L);
    const bool valid = trial < A.T;
    if (!valid) trial = A.T - 1;
    const uvs_filter_params &fp = A.fp;
    Rows<M, N, L> st;
    double z[R], err[R], kap[R], h[N], cmd[N];
#pragma unroll
    for (int j = 0; j < N; ++j) h[j] = A.first ? 0.0 : A.dq_prev[trial * N + j];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = sub * R + r;
        const double fv = A.f[trial * M + row];
        z[r] = fv - A.f_old[trial * M + row];
        err[r] = fv - fp.desired[row];
#pragma unroll
        for (int j = 0; j < N; ++j) st.x[r][j] = A.X[(trial * M + row) * N + j];
#pragma unroll
        for (int l = 0; l < N; ++l)
#pragma unroll
            for (int j = l; j < N; ++j) st.p[r][Sym<N>::at(l, j)] = A.P[((trial * M + row) * N + l) * N + j];
    }
    st.template update<METHOD_T>(fp, z, h, bandwidth(fp, A.k), kap);
    const int bad = st.any_nonfinite();
    control_law<M, N, L>(st, kap, err, fp.gain, sub, cmd);
    if (!valid) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = sub * R + r;
        A.err_out[trial * M + row] = err[r];
        A.kappa_out[trial * M + row] = kap[r];
#pragma unroll
        for (int j = 0; j < N; ++j) A.X[(trial * M + row) * N + j] = st.x[r][j];
#pragma unroll
        for (int l = 0; l < N; ++l)
#pragma unroll
            for (int j = 0; j < N; ++j) A.P[((trial * M + row) * N + l) * N + j] = st.p[r][Sym<N>::at(l, j)];
    }
    if (sub == 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) A.dq_out[trial * N + j] = cmd[j];
        A.status[trial] = bad ? UVS_STATUS_FAIL : UVS_STATUS_SUCCESS;
    }
}

// ------------------------------------------------------------------------------------------------ statistics
// One lane per (trial, feature) pair would under-fill short batches; one lane per trial reading m strided columns
// keeps the trial-fastest layout coalesced.
__global__ __launch_bounds__(256) void stats_kernel(long long T, int K, int m, View err, const double *t, const int *k_done, double *stats) {
    const long long trial = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (trial >= T) return;
    const int rows = k_done ? k_done[trial] : K;
    double n_ise = 0.0, n_iae = 0.0, n_itae = 0.0;
    for (int i = 0; i < m; ++i) {
        double ise = 0.0, iae = 0.0, itae = 0.0;
        for (int k = 0; k < rows; ++k) {
            const double e = *err.at(trial, k, i), ae = fabs(e);
            ise = fma(e, e, ise);
            iae += ae;
            itae = fma(t[k], ae, itae);
        }
        n_ise = fma(ise, ise, n_ise);
        n_iae = fma(iae, iae, n_iae);
        n_itae = fma(itae, itae, n_itae);
    }
    stats[3 * trial] = sqrt(n_ise);
    stats[3 * trial + 1] = sqrt(n_iae);
    stats[3 * trial + 2] = sqrt(n_itae);
}

// ------------------------------------------------------------------------------------------------ math self-test
__global__ __launch_bounds__(256) void debug_math_kernel(int which, long long n, const double *x, double *y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double s, c, r;
    switch (which) {
        case 0: y[i] = fast_rcp(v); break;
        case 1: fast_sqrt_rsqrt(v, s, r); y[i] = s; break;
        case 2: fast_sqrt_rsqrt(v, s, r); y[i] = r; break;
        case 3: sincos_any(v, s, c); y[i] = s; break;
        case 4: sincos_any(v, s, c); y[i] = c; break;
        case 6: y[i] = exp_nonpos(v); break;
        default: y[i] = exp(v); break;
    }
}

}  // namespace uvs

// ================================================================================================ C ABI
namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    std::snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
        return UVS_ERR_HIP;
    }
    return UVS_OK;
}

// (m, n, lanes-per-filter) instantiations; the first listed L of a shape is its default.
#ifdef UVS_QUICK                      // experiment builds (make quick): the headline shape only, compiles in seconds
#define UVS_SHAPES(X) X(8, 6, 2)
#define UVS_TUNED_SHAPES(X) X(8, 6, 2)
#else
#define UVS_SHAPES(X) \
    X(8, 6, 2) X(8, 6, 1) X(8, 6, 4) X(8, 6, 8) \
    X(2, 6, 1) \
    X(6, 6, 2) X(6, 6, 1) \
    X(32, 7, 16) X(32, 7, 32) X(32, 7, 8)

#define UVS_TUNED_SHAPES(X) X(8, 6, 1) X(8, 6, 2) X(8, 6, 4) X(6, 6, 2)
#endif

int default_lanes(int m, int n, int method) {
    // MCKF carries the Cholesky factors of its blocks through the fixed-point passes: at two lanes per filter (4 blocks per lane) that
    // state goes to scratch (31 ms per 65 536 x 299 sweep), at four lanes it stays in registers (8 ms)
    if (method == UVS_METHOD_MCKF && m == 8 && n == 6) return 4;
#define X(M, N, L) if (m == M && n == N) return L;
    UVS_SHAPES(X)
#undef X
    return 0;
}

int check_params(const uvs_filter_params *fp, int64_t T, int *lanes) {
    if (!fp) return fail(UVS_ERR_ARG, "%s", "filter params are NULL");
    if (T <= 0) return fail(UVS_ERR_ARG, "%s", "T must be positive");
    if (fp->steps < 0 || fp->k_max <= 0) return fail(UVS_ERR_ARG, "%s", "steps must be >= 0 and k_max > 0");
    if (fp->method != UVS_METHOD_KF && fp->method != UVS_METHOD_MCKF && fp->method != UVS_METHOD_IMCCKF && fp->method != UVS_METHOD_GMCKF)
        return fail(UVS_ERR_METHOD, "%s", "method must be KF, MCKF, IMCCKF or GMCKF");
    if (fp->method == UVS_METHOD_MCKF && fp->fpi_epoch_max < 1) return fail(UVS_ERR_ARG, "%s", "MCKF needs fpi_epoch_max >= 1");
    const int L = fp->lanes_per_filter < 0 ? -fp->lanes_per_filter : (fp->lanes_per_filter ? fp->lanes_per_filter : default_lanes(fp->m, fp->n, fp->method));
    if (L == 0) return fail(UVS_ERR_SHAPE, "%s", "(m, n) is not instantiated in libuvs_rmckf");
    *lanes = L;
    return UVS_OK;
}

dim3 grid_for(int64_t T, int L) { return dim3((unsigned)((T * L + 63) / 64)); }

// Tuned closed-loop kernel: estimator, plant kind and "X stream wanted" are compile-time there.
template <int M, int N, int LL, int METHOD, int PLANT>
void launch_tuned2(bool xo, dim3 g, hipStream_t s, const uvs::ClosedArgs &A) {
#ifdef UVS_PV                          // experiment builds: number of covariance blocks per lane kept in registers at L = 2
    constexpr int PV = (LL == 2 ? UVS_PV : M / LL);
#else
    constexpr int PV = (LL == 2 ? M / LL / 2 : M / LL);      // L = 2 parks half of its blocks in LDS; 1 and 4 keep all in registers
#endif
    if (xo) hipLaunchKernelGGL((uvs::closed_loop_tuned_kernel<M, N, LL, METHOD, PLANT, PV, true>), g, dim3(64), 0, s, A);
    else hipLaunchKernelGGL((uvs::closed_loop_tuned_kernel<M, N, LL, METHOD, PLANT, PV, false>), g, dim3(64), 0, s, A);
}
template <int M, int N, int LL>
void launch_tuned(int method, bool linear, bool xo, dim3 g, hipStream_t s, const uvs::ClosedArgs &A) {
    if (method == UVS_METHOD_GMCKF && !linear) launch_tuned2<M, N, LL, UVS_METHOD_GMCKF, UVS_PLANT_DH_PINHOLE>(xo, g, s, A);
    else if (method == UVS_METHOD_GMCKF) launch_tuned2<M, N, LL, UVS_METHOD_GMCKF, UVS_PLANT_LINEAR>(xo, g, s, A);
    else if (method == UVS_METHOD_IMCCKF && !linear) launch_tuned2<M, N, LL, UVS_METHOD_IMCCKF, UVS_PLANT_DH_PINHOLE>(xo, g, s, A);
    else if (method == UVS_METHOD_IMCCKF) launch_tuned2<M, N, LL, UVS_METHOD_IMCCKF, UVS_PLANT_LINEAR>(xo, g, s, A);
    else if (!linear) launch_tuned2<M, N, LL, UVS_METHOD_KF, UVS_PLANT_DH_PINHOLE>(xo, g, s, A);
    else launch_tuned2<M, N, LL, UVS_METHOD_KF, UVS_PLANT_LINEAR>(xo, g, s, A);
}

// Tuned replay kernel (rmckf_replay_tuned.hpp): estimator, "X stream wanted" and "control law wanted" are compile-time.
#ifndef UVS_REPLAY_PV
#define UVS_REPLAY_PV 2
#endif
#define UVS_TUNED_REPLAY_SHAPES(X) X(8, 6) X(6, 6)
template <int M, int N, int METHOD>
void launch_replay_tuned2(bool xo, bool cmd, dim3 g, hipStream_t s, const uvs::ReplayArgs &A) {
    constexpr int PV = UVS_REPLAY_PV;
    if (xo && cmd) hipLaunchKernelGGL((uvs::replay_tuned_kernel<M, N, METHOD, PV, true, true>), g, dim3(64), 0, s, A);
    else if (xo) hipLaunchKernelGGL((uvs::replay_tuned_kernel<M, N, METHOD, PV, true, false>), g, dim3(64), 0, s, A);
    else if (cmd) hipLaunchKernelGGL((uvs::replay_tuned_kernel<M, N, METHOD, PV, false, true>), g, dim3(64), 0, s, A);
    else hipLaunchKernelGGL((uvs::replay_tuned_kernel<M, N, METHOD, PV, false, false>), g, dim3(64), 0, s, A);
}
template <int M, int N>
void launch_replay_tuned(int method, bool xo, bool cmd, dim3 g, hipStream_t s, const uvs::ReplayArgs &A) {
    if (method == UVS_METHOD_GMCKF) launch_replay_tuned2<M, N, UVS_METHOD_GMCKF>(xo, cmd, g, s, A);
    else if (method == UVS_METHOD_IMCCKF) launch_replay_tuned2<M, N, UVS_METHOD_IMCCKF>(xo, cmd, g, s, A);
    else launch_replay_tuned2<M, N, UVS_METHOD_KF>(xo, cmd, g, s, A);
}

// Estimator-only replay, four lanes per filter, state in registers, two wavefronts per SIMD (rmckf_replay_tuned.hpp).
template <int M, int N>
void launch_replay_rows(int method, bool xo, bool eo, int64_t T, hipStream_t s, const uvs::ReplayArgs &A) {
    const dim3 g = grid_for(T, 4), b(64);
#define UVS_ROWS(METHOD) \
    if (xo && eo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, true, true>), g, b, 0, s, A); \
    else if (xo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, true, false>), g, b, 0, s, A); \
    else if (eo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, false, true>), g, b, 0, s, A); \
    else hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, false, false>), g, b, 0, s, A);
    if (method == UVS_METHOD_GMCKF) { UVS_ROWS(UVS_METHOD_GMCKF) }
    else if (method == UVS_METHOD_IMCCKF) { UVS_ROWS(UVS_METHOD_IMCCKF) }
    else { UVS_ROWS(UVS_METHOD_KF) }
#undef UVS_ROWS
}

}  // namespace

extern "C" {

const char *uvs_version(void) { return "uvs_rmckf 0.1.0 (gfx950, fp64)"; }
const char *uvs_last_error(void) { return g_err; }

int uvs_supported_lanes(int32_t m, int32_t n, int32_t *lanes, int32_t cap) {
    int cnt = 0;
#define X(M, N, L) if (m == M && n == N) { if (lanes && cnt < cap) lanes[cnt] = L; ++cnt; }
    UVS_SHAPES(X)
#undef X
    return cnt;
}

int uvs_rmckf_closed_loop_f64(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T, uvs_view q_start, uvs_view noise,
                              uvs_view x0, uvs_view x_out, uvs_view err_out, uvs_view q_out, uvs_view f_out, uvs_view dq_out,
                              double *stats, int32_t *status, int32_t *k_done, uvs_view x_final, uvs_view p_final, void *stream) {
    int L = 0;
    if (int rc = check_params(fp, T, &L)) return rc;
    if (!plant) return fail(UVS_ERR_ARG, "%s", "plant is NULL");
    if (plant->n_joints != fp->n) return fail(UVS_ERR_ARG, "%s", "plant does not match n");
    if (plant->kind == UVS_PLANT_DH_PINHOLE && plant->n_points * 2 != fp->m) return fail(UVS_ERR_ARG, "%s", "plant does not match m");
    if (plant->kind == UVS_PLANT_LINEAR) {
        if (!plant->lin_jacobian || !plant->lin_f0 || !plant->lin_q0) return fail(UVS_ERR_ARG, "%s", "linear plant arrays are NULL");
        if (fp->initial_guess) return fail(UVS_ERR_ARG, "%s", "the analytic initial guess needs the DH/pinhole plant; pass x0");
    } else if (plant->kind != UVS_PLANT_DH_PINHOLE) {
        return fail(UVS_ERR_ARG, "%s", "unknown plant kind");
    }
    if (!q_start.base) return fail(UVS_ERR_ARG, "%s", "q_start view is NULL");
    if (!fp->initial_guess && !x0.base) return fail(UVS_ERR_ARG, "%s", "x0 view is required when initial_guess == 0");
    uvs::ClosedArgs A;
    A.fp = *fp;
    A.plant = *plant;
    A.T = T;
    A.q_start = uvs::to_view(q_start); A.noise = uvs::to_view(noise); A.x0 = uvs::to_view(x0);
    A.x_out = uvs::to_view(x_out); A.err_out = uvs::to_view(err_out); A.q_out = uvs::to_view(q_out);
    A.f_out = uvs::to_view(f_out); A.dq_out = uvs::to_view(dq_out);
    A.x_final = uvs::to_view(x_final); A.p_final = uvs::to_view(p_final);
    A.stats = stats; A.status = status; A.k_done = k_done;
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
    // lanes_per_filter 1 / 2 / 4 select the tuned kernel (rmckf_tuned.hpp) where it exists; a negative value forces the generic
    // template with |value| lanes (kept as an in-library cross-check of the tuned code).
    const bool tuned_ok = (fp->method == UVS_METHOD_GMCKF || fp->method == UVS_METHOD_KF || fp->method == UVS_METHOD_IMCCKF) && fp->lanes_per_filter >= 0;
#define XT(M, N, LL) \
    if (!launched && tuned_ok && L == LL && fp->m == M && fp->n == N) { \
        launch_tuned<M, N, LL>(fp->method, plant->kind == UVS_PLANT_LINEAR, x_out.base != nullptr, grid_for(T, LL), s, A); \
        launched = true; \
    }
    UVS_TUNED_SHAPES(XT)
#undef XT
#define X(M, N, LL) \
    if (!launched && fp->m == M && fp->n == N && L == LL) { \
        if (fp->method == UVS_METHOD_GMCKF) hipLaunchKernelGGL((uvs::closed_loop_kernel<M, N, LL, UVS_METHOD_GMCKF>), grid_for(T, LL), dim3(64), 0, s, A); \
        else hipLaunchKernelGGL((uvs::closed_loop_kernel<M, N, LL, 0>), grid_for(T, LL), dim3(64), 0, s, A); \
        launched = true; \
    }
    UVS_SHAPES(X)
#undef X
    if (!launched) return fail(UVS_ERR_SHAPE, "%s", "(m, n, lanes_per_filter) is not instantiated in libuvs_rmckf");
    return check_launch("closed_loop_kernel");
}

int uvs_rmckf_replay_f64(const uvs_filter_params *fp, int64_t T, uvs_view f, uvs_view dq, uvs_view x0, uvs_view x_out,
                         uvs_view err_out, uvs_view kappa_out, uvs_view dqcmd_out, int32_t *status, int32_t *k_done,
                         uvs_view x_final, uvs_view p_final, void *stream) {
    int L = 0;
    if (int rc = check_params(fp, T, &L)) return rc;
    if (!f.base || !dq.base || !x0.base) return fail(UVS_ERR_ARG, "%s", "f, dq and x0 views are required");
    uvs::ReplayArgs A;
    A.fp = *fp;
    A.T = T;
    A.f = uvs::to_view(f); A.dq = uvs::to_view(dq); A.x0 = uvs::to_view(x0);
    A.x_out = uvs::to_view(x_out); A.err_out = uvs::to_view(err_out); A.kappa_out = uvs::to_view(kappa_out);
    A.dqcmd_out = uvs::to_view(dqcmd_out); A.x_final = uvs::to_view(x_final); A.p_final = uvs::to_view(p_final);
    A.status = status; A.k_done = k_done;
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
    // two lanes per filter (the default) at (8,6): tuned kernel; a negative lanes_per_filter forces the generic template
    const bool tuned_method = fp->method == UVS_METHOD_GMCKF || fp->method == UVS_METHOD_KF || fp->method == UVS_METHOD_IMCCKF;
    const bool tuned_ok = tuned_method && fp->lanes_per_filter >= 0 && L == 2;
    // without the commanded dq there is no least-squares solve and nothing couples a filter's rows: four lanes per filter, state in
    // registers, two wavefronts per SIMD (library default, or lanes_per_filter = 4)
    if (tuned_method && !dqcmd_out.base && fp->m == 8 && fp->n == 6 && (fp->lanes_per_filter == 0 || fp->lanes_per_filter == 4)) {
        launch_replay_rows<8, 6>(fp->method, x_out.base != nullptr, err_out.base != nullptr, T, s, A);
        launched = true;
    }
#define XR(M, N) \
    if (!launched && tuned_ok && fp->m == M && fp->n == N) { \
        launch_replay_tuned<M, N>(fp->method, x_out.base != nullptr, dqcmd_out.base != nullptr, grid_for(T, 2), s, A); \
        launched = true; \
    }
    UVS_TUNED_REPLAY_SHAPES(XR)
#undef XR
#define X(M, N, LL) \
    if (!launched && fp->m == M && fp->n == N && L == LL) { \
        if (fp->method == UVS_METHOD_GMCKF) hipLaunchKernelGGL((uvs::replay_kernel<M, N, LL, UVS_METHOD_GMCKF>), grid_for(T, LL), dim3(64), 0, s, A); \
        else hipLaunchKernelGGL((uvs::replay_kernel<M, N, LL, 0>), grid_for(T, LL), dim3(64), 0, s, A); \
        launched = true; \
    }
    UVS_SHAPES(X)
#undef X
    if (!launched) return fail(UVS_ERR_SHAPE, "%s", "(m, n, lanes_per_filter) is not instantiated in libuvs_rmckf");
    return check_launch("replay_kernel");
}

int uvs_rmckf_step_f64(const uvs_filter_params *fp, int64_t T, double *X, double *P, const double *f, const double *f_old,
                       const double *dq_prev, int32_t first, int32_t k, double *dq_out, double *err_out, double *kappa_out,
                       int32_t *status, void *stream) {
    int L = 0;
    if (int rc = check_params(fp, T, &L)) return rc;
    if (!X || !P || !f || !f_old || !dq_prev || !dq_out || !err_out || !kappa_out || !status)
        return fail(UVS_ERR_ARG, "%s", "all step buffers are required");
    uvs::StepArgs A{*fp, T, X, P, f, f_old, dq_prev, first, k, dq_out, err_out, kappa_out, status};
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
#define X(M, N, LL) \
    if (!launched && fp->m == M && fp->n == N && L == LL) { \
        hipLaunchKernelGGL((uvs::step_kernel<M, N, LL, 0>), grid_for(T, LL), dim3(64), 0, s, A); \
        launched = true; \
    }
    UVS_SHAPES(X)
#undef X
    if (!launched) return fail(UVS_ERR_SHAPE, "%s", "(m, n, lanes_per_filter) is not instantiated in libuvs_rmckf");
    return check_launch("step_kernel");
}

int uvs_stats_reduce_f64(int64_t T, int32_t K, int32_t m, uvs_view err, const double *t, const int32_t *k_done, double *stats,
                         void *stream) {
    if (T <= 0 || K < 0 || m <= 0 || !err.base || !t || !stats) return fail(UVS_ERR_ARG, "%s", "bad stats arguments");
    hipLaunchKernelGGL(uvs::stats_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)T, K, m,
                       uvs::to_view(err), t, k_done, stats);
    return check_launch("stats_kernel");
}

int uvs_noise_generate_f64(const uvs_noise_params *np, int64_t T, const uint64_t *states, const double *zig, uvs_view out, void *stream) {
    if (!np || T <= 0 || !states || !zig || !out.base) return fail(UVS_ERR_ARG, "%s", "bad noise_generate arguments");
    if (np->m <= 0 || np->m % 2 || np->m > UVS_MAX_M || np->steps < 0) return fail(UVS_ERR_ARG, "%s", "noise: m must be even and <= UVS_MAX_M");
    if (np->type < UVS_NOISE_WHITE || np->type > UVS_NOISE_UNIFORM) return fail(UVS_ERR_ARG, "%s", "unknown noise type");
    uvs::NoiseArgs A{*np, (long long)T, (const unsigned long long *)states, zig, uvs::to_view(out)};
    const long long lanes = (long long)T * (np->m / 2);
    hipLaunchKernelGGL(uvs::noise_kernel, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, (hipStream_t)stream, A);
    return check_launch("noise_kernel");
}

int uvs_pcg64_seed_u64(int64_t n, const uint64_t *seeds, uint64_t *states, void *stream) {
    if (n <= 0 || !seeds || !states) return fail(UVS_ERR_ARG, "%s", "bad pcg64_seed arguments");
    hipLaunchKernelGGL(uvs::pcg64_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n,
                       (const unsigned long long *)seeds, (unsigned long long *)states);
    return check_launch("pcg64_seed_kernel");
}

int uvs_debug_math_f64(int32_t which, int64_t n, const double *x, double *y, void *stream) {
    if (n <= 0 || !x || !y) return fail(UVS_ERR_ARG, "%s", "bad debug_math arguments");
    hipLaunchKernelGGL(uvs::debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, which, (long long)n, x, y);
    return check_launch("debug_math_kernel");
}

}  // extern "C"

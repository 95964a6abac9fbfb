// Generic kernel templates: any (m, n) and any power-of-two lanes per filter, every estimator (METHOD_T = 0 selects at run time, which
// is also the only home of the fixed-point MCKF).  The tuned kernels (rmckf_tuned.hpp, rmckf_replay_tuned.hpp) cover the headline shapes;
// these cover the rest and serve as the in-library cross-check of the tuned code (negative lanes_per_filter).
#pragma once
#include <hip/hip_runtime.h>
#include "rmckf_device.hpp"

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
// CAREFUL = true is the second pass behind any closed-loop kernel: it re-runs, from their first step, exactly the trials the first pass
// marked UVS_STATUS_SUSPECT (a numerically rank-deficient Jacobian showed up in the control law) with numpy's pinv semantics, and
// overwrites their outputs; wavefronts without such a trial exit at once.
template <int M, int N, int L, int METHOD_T, bool CAREFUL = false>
__global__ __launch_bounds__(64) void closed_loop_kernel(const ClosedArgs A) {
    constexpr int R = M / L;
    const long long gl = (long long)blockIdx.x * 64 + threadIdx.x;
    long long trial = gl / L;
    const int sub = (int)(gl % L);
    const bool valid = trial < A.T;
    if (!valid) trial = A.T - 1;                    // padding lanes shadow the last trial so group shuffles stay uniform
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;
    bool mine = true, flagged = false;
    if constexpr (CAREFUL) {
        mine = A.status[trial] == UVS_STATUS_SUSPECT;
        if (!__any(mine)) return;
    }

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
        const bool suspect = control_law<M, N, L, CAREFUL>(st, kap, err, fp.gain, sub, dq);
        flagged |= alive && suspect;

        if (alive && valid && mine) {
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
    if (valid && mine) {
        store_final<M, N, L>(st, A.x_final, A.p_final, trial, sub);
        if (sub == 0) {
            if (A.stats) { A.stats[3 * trial] = s2[0]; A.stats[3 * trial + 1] = s2[1]; A.stats[3 * trial + 2] = s2[2]; }
            if (A.status) A.status[trial] = (!CAREFUL && flagged) ? UVS_STATUS_SUSPECT : status;
            if (A.k_done) A.k_done[trial] = k_done;
        }
    }
}

// ------------------------------------------------------------------------------------------------ replay
template <int M, int N, int L, int METHOD_T, bool CAREFUL = false>
__global__ __launch_bounds__(64) void replay_kernel(const ReplayArgs A) {
    constexpr int R = M / L;
    const long long gl = (long long)blockIdx.x * 64 + threadIdx.x;
    long long trial = gl / L;
    const int sub = (int)(gl % L);
    const bool valid = trial < A.T;
    if (!valid) trial = A.T - 1;
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;
    bool mine = true, flagged = false;
    if constexpr (CAREFUL) {                        // second pass: only the trials the first pass marked suspect (see closed_loop_kernel)
        mine = A.status[trial] == UVS_STATUS_SUSPECT;
        if (!__any(mine)) return;
    }

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
        const bool suspect = control_law<M, N, L, CAREFUL>(st, kap, err, fp.gain, sub, cmd);
        flagged |= alive && suspect && A.dqcmd_out.on();        // the careful pass only follows when the command is an output
        if (alive && valid && mine) {
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
    if (valid && mine) {
        store_final<M, N, L>(st, A.x_final, A.p_final, trial, sub);
        if (sub == 0) {
            if (A.status) A.status[trial] = (!CAREFUL && flagged) ? UVS_STATUS_SUSPECT : status;
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
    // numpy's pinv semantics inline (there is no second pass behind a single step).  Round 5 ran the careful solve -- Householder QR finished by a Jacobi
    // SVD of the factor -- on every call: 45 of the 59 us of a drop-in step were this one wavefront's SVD sweeps (tools/time_step_route.py).  Round 6: the
    // plain QR with the default mode's watches first (spread of the factor's entries, growth of the solution), the careful solve only for a filter whose
    // watch fires -- or for every filter under UVS_OPT_STRICT_PINV.  A filter's command never depends on its neighbours in the wavefront.
    const bool strict = (fp.reserved & UVS_OPT_STRICT_PINV) != 0;
    const bool suspect = strict || control_law<M, N, L, false>(st, kap, err, fp.gain, sub, cmd);
    if (__any(suspect)) {
        double careful[N];
        control_law<M, N, L, true>(st, kap, err, fp.gain, sub, careful);
#pragma unroll
        for (int j = 0; j < N; ++j) cmd[j] = suspect ? careful[j] : cmd[j];
    }
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

}  // namespace uvs

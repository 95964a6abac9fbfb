// Estimator-only replay in single precision -- the "measured variant" of SURVEY.md sections 0 (fact 5), 7 (step 2) and 8d: the same
// row-form estimator as experiment.py:166-297 over recorded feature / joint-delta streams, fp32 streams AND fp32 state.  It is never the
// headline and carries no parity gate beyond a measured bound (tests/test_gpu_replay_f32.py prints the per-fixture error against the
// reference's fp64 runs): its purpose is to show what the streaming design does once VALU issue stops binding.  fp64 arithmetic issues one
// wave64 FMA per 4 cycles and binds every fp64 kernel of this library; here two rows of a filter ride in one register pair and every
// multiply-add is a v_pk_fma_f32 (two fp32 FMAs per lane and instruction, same issue cost), so the arithmetic of a step shrinks 4x per
// trial and the kernel is left with its streams: read f (m) + dq (n), write X (mn) + err (m) = 4 (2m + n + mn) = 280 B per update at (8,6).
//
// Mapping: two lanes per filter, lane `sub` owns rows 4 sub .. 4 sub + 3 as two row pairs; P (2 x 21 pairs), X (2 x 6 pairs) in VGPRs,
// no LDS, no cross-lane traffic except IMCC-KF's one innovation norm; two wavefronts per SIMD (at three the register cap of 168 spills).
// Trial-fastest streams give 128-byte segments per store instruction and lane group.
#pragma once
#include <hip/hip_runtime.h>
#include "rmckf_device.hpp"

namespace uvs {

typedef float v2f __attribute__((ext_vector_type(2)));

struct View32 {
    float *p;
    long long st, sk, sc;
    UVS_DEV float *at(long long t, long long k, long long c) const { return p + t * st + k * sk + c * sc; }
};

struct ReplayArgs32 {
    uvs_filter_params fp;
    long long T;
    View32 f, dq, x0, x_out, err_out;
    int *status, *k_done;
};

UVS_DEV v2f splat(float v) { return v2f{v, v}; }
UVS_DEV v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// 1 / v to fp32 rounding: v_rcp_f32 (1 ulp) + one Newton step
UVS_DEV float rcp32(float v) {
    const float r = __builtin_amdgcn_rcpf(v);
    return fmaf(r, fmaf(-v, r, 1.0f), r);
}

template <int METHOD, bool XOUT, bool EOUT>
__global__ __launch_bounds__(64, 2) void replay_f32_kernel(const ReplayArgs32 A) {
    constexpr int M = 8, N = 6, L = 2, RP = M / L / 2, NP = Sym<N>::NP, TPW = 64 / L;     // RP row pairs per lane
    const unsigned lane = threadIdx.x;
    // blocked lane mapping: lanes 0-31 hold rows 0-3 of 32 consecutive trials, lanes 32-63 rows 4-7 -- every store instruction writes two
    // whole 128-byte segments (partner lanes side by side would interleave two address streams lane by lane: measured 1.25x slower on the
    // fp64 replay kernels)
    const int sub = (int)(lane >> 5);
    const unsigned tl = lane & 31u;
    const long long wave_first = (long long)blockIdx.x * TPW;
    const bool valid = wave_first + tl < A.T;
    const long long trial = valid ? wave_first + tl : A.T - 1;                // padding lanes shadow the last trial
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;
    const int row0 = sub * 2 * RP;                                            // first row of this lane

    v2f x[RP][N], p[RP][NP], f_prev[RP], des[RP];
#pragma unroll
    for (int q = 0; q < RP; ++q) {
        const int r = row0 + 2 * q;
        des[q] = v2f{(float)fp.desired[r], (float)fp.desired[r + 1]};
        f_prev[q] = v2f{*A.f.at(trial, 0, r), *A.f.at(trial, 0, r + 1)};
#pragma unroll
        for (int j = 0; j < N; ++j) x[q][j] = v2f{*A.x0.at(trial, 0, r * N + j), *A.x0.at(trial, 0, (r + 1) * N + j)};
#pragma unroll
        for (int l = 0; l < N; ++l)
#pragma unroll
            for (int j = l; j < N; ++j) p[q][Sym<N>::at(l, j)] = splat(l == j ? 1.0f : 0.0f);      // P = I (experiment.py:73)
    }
    const float *pf = A.f.at(trial, 1, row0);
    const float *pd = A.dq.at(trial, 1, 0);
    float *px = (XOUT && A.x_out.p) ? A.x_out.at(trial, 0, row0 * N) : nullptr;
    float *pe = (EOUT && A.err_out.p) ? A.err_out.at(trial, 0, row0) : nullptr;

    v2f f_next[RP];
    float h_next[N];
#pragma unroll
    for (int q = 0; q < RP; ++q) f_next[q] = splat(0.0f);
#pragma unroll
    for (int j = 0; j < N; ++j) h_next[j] = 0.0f;                            // first_run: H = 0 (experiment.py:183-185)
    if (K > 0) {
#pragma unroll
        for (int q = 0; q < RP; ++q) f_next[q] = v2f{pf[(2 * q) * A.f.sc], pf[(2 * q + 1) * A.f.sc]};
        pf += A.f.sk;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                      // vmcnt(0), see rmckf_replay_tuned.hpp

    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true;
    const float reg = (float)fp.reg;
    for (int k = 0; k < K; ++k) {
        v2f f[RP];
        float h[N];
#pragma unroll
        for (int q = 0; q < RP; ++q) f[q] = f_next[q];
#pragma unroll
        for (int j = 0; j < N; ++j) h[j] = h_next[j];
        if (k + 1 < K) {                                                     // inputs of step k + 1: a whole step to arrive
#pragma unroll
            for (int q = 0; q < RP; ++q) f_next[q] = v2f{pf[(2 * q) * A.f.sc], pf[(2 * q + 1) * A.f.sc]};
#pragma unroll
            for (int j = 0; j < N; ++j) h_next[j] = pd[j * A.dq.sc];
            pf += A.f.sk;
            pd += A.dq.sk;
        }
        const float sigma = (float)bandwidth(fp, k);
        const float c_exp2 = -0.5f * 1.44269504088896341f * rcp32(sigma * sigma);   // exp(-nu^2 / (2 sigma^2)) = exp2(nu^2 c)
        v2f nu[RP];
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            v2f pred = splat(0.0f);
#pragma unroll
            for (int j = 0; j < N; ++j) pred = pk_fma(x[q][j], splat(h[j]), pred);
            nu[q] = (f[q] - f_prev[q]) - pred;                               // innovation (experiment.py:170-177, 274)
            f_prev[q] = f[q];
        }
        float c_shared = 1.0f;
        if constexpr (METHOD == UVS_METHOD_IMCCKF) {                         // one weight for the whole filter (experiment.py:258-261)
            float ss = 0.0f;
#pragma unroll
            for (int q = 0; q < RP; ++q) ss = fmaf(nu[q].x, nu[q].x, fmaf(nu[q].y, nu[q].y, ss));
            ss += __shfl_xor(ss, 32, 64);
            c_shared = __builtin_amdgcn_exp2f(ss * c_exp2);
        }
        v2f chk = splat(0.0f);                                               // turns NaN as soon as a state entry is non-finite
        float *pxc = px, *pec = pe;
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            v2f g[N];
#pragma unroll
            for (int l = 0; l < N; ++l) p[q][Sym<N>::at(l, l)] += splat(1.0f);                 // P + Q (experiment.py:167)
#pragma unroll
            for (int l = 0; l < N; ++l) {
                v2f acc = p[q][Sym<N>::at(l, 0)] * splat(h[0]);
#pragma unroll
                for (int j = 1; j < N; ++j) acc = pk_fma(p[q][Sym<N>::at(l, j)], splat(h[j]), acc);
                g[l] = acc;
            }
            v2f a = splat(0.0f);
#pragma unroll
            for (int l = 0; l < N; ++l) a = pk_fma(splat(h[l]), g[l], a);
            v2f gamma;
            if constexpr (METHOD == UVS_METHOD_GMCKF) {                      // utils.py:171-172, experiment.py:280-286
                const v2f arg = (nu[q] * nu[q]) * splat(c_exp2);
                const v2f d = v2f{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)} + splat(reg);
                const v2f den = pk_fma(a, d, splat(1.0f));
                gamma = d * v2f{rcp32(den.x), rcp32(den.y)};
            } else if constexpr (METHOD == UVS_METHOD_IMCCKF) {              // experiment.py:262-264
                const v2f den = pk_fma(splat(c_shared), a, splat(1.0f));
                gamma = splat(c_shared) * v2f{rcp32(den.x), rcp32(den.y)};
            } else {                                                         // KF (experiment.py:192)
                const v2f den = a + splat(1.0f);
                gamma = v2f{rcp32(den.x), rcp32(den.y)};
            }
            const v2f step = gamma * nu[q];
            const v2f beta = gamma * (splat(2.0f) - gamma * (a + splat(1.0f)));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                x[q][j] = pk_fma(g[j], step, x[q][j]);                       // X + K (Z - H X) (experiment.py:291)
                chk = pk_fma(x[q][j], splat(0.0f), chk);
            }
#pragma unroll
            for (int l = 0; l < N; ++l) {                                    // Joseph update with R = 1: P -= beta g g^T (experiment.py:296-297)
                const v2f w = -(beta * g[l]);
#pragma unroll
                for (int j = l; j < N; ++j) p[q][Sym<N>::at(l, j)] = pk_fma(w, g[j], p[q][Sym<N>::at(l, j)]);
            }
            if constexpr (XOUT) {                                            // components of the pair's two rows are consecutive: one running pointer
#pragma unroll
                for (int j = 0; j < N; ++j) { *pxc = x[q][j].x; pxc += A.x_out.sc; }
#pragma unroll
                for (int j = 0; j < N; ++j) { *pxc = x[q][j].y; pxc += A.x_out.sc; }
            }
            if constexpr (EOUT) {
                const v2f e = f[q] - des[q];                                 // experiment.py:302
                *pec = e.x; pec += A.err_out.sc;
                *pec = e.y; pec += A.err_out.sc;
            }
        }
        if constexpr (XOUT) px += A.x_out.sk;
        if constexpr (EOUT) pe += A.err_out.sk;
        float bad = chk.x + chk.y;
        bad += __shfl_xor(bad, 32, 64);
        if (alive && !(bad == 0.0f)) {                                       // X non-finite: pinv would raise in the control law (experiment.py:313-316)
            alive = false;
            status = UVS_STATUS_FAIL;
            k_done = k;
        }
        if (!__any(alive)) break;
    }
    if (valid && sub == 0) {
        if (A.status) A.status[trial] = status;
        if (A.k_done) A.k_done[trial] = k_done;
    }
}

}  // namespace uvs

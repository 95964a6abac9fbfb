// Non-template kernels: error statistics of a logged trajectory and the math test hook.  Included by exactly one translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include "rmckf_device.hpp"

namespace uvs {

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
        case 7: y[i] = log_any(v); break;
        case 8: y[i] = exp_clamped(v); break;
        // 9-12: the tracked (sin, cos) pair of the angle 0.7 rotated by v -- short polynomials (|v| <= 0.1), long ones (|v| <= 1)
        case 9: s = 0.64421768723769102; c = 0.76484218728448850; sincos_advance(s, c, v); y[i] = s; break;
        case 10: s = 0.64421768723769102; c = 0.76484218728448850; sincos_advance(s, c, v); y[i] = c; break;
        case 11: s = 0.64421768723769102; c = 0.76484218728448850; sincos_advance_wide(s, c, v); y[i] = s; break;
        case 12: s = 0.64421768723769102; c = 0.76484218728448850; sincos_advance_wide(s, c, v); y[i] = c; break;
        default: y[i] = exp(v); break;
    }
}

}  // namespace uvs

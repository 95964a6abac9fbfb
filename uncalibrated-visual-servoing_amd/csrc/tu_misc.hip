// Single-step kernel (state in HBM), statistics, math test hook, noise generator and PCG64 seeding.
#include "launchers.hpp"
#include "rmckf_generic.hpp"
#include "misc_kernels.hpp"
#include "noise_kernels.hpp"

bool uvs_launch::step_generic(int m, int n, int L, int64_t T, hipStream_t s, const uvs::StepArgs &A) {
#define X(M, N, LL) \
    if (m == M && n == N && L == LL) { hipLaunchKernelGGL((uvs::step_kernel<M, N, LL, 0>), grid_for(T, LL), dim3(64), 0, s, A); return true; }
    UVS_SHAPES(X)
#undef X
    return false;
}

namespace uvs {
__global__ void fill_i32_kernel(int *p, int v, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
}  // namespace uvs
void uvs_launch::fill_i32(int *p, int v, long long n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(uvs::fill_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
}
void uvs_launch::stats(long long T, int K, int m, uvs::View err, const double *t, const int *k_done, double *out, hipStream_t s) {
    hipLaunchKernelGGL(uvs::stats_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, T, K, m, err, t, k_done, out);
}

void uvs_launch::debug_math(int which, long long n, const double *x, double *y, hipStream_t s) {
    hipLaunchKernelGGL(uvs::debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, which, n, x, y);
}

void uvs_launch::noise(const uvs_noise_params &np_in, long long T, const unsigned long long *states, const double *zig, uvs::View out, hipStream_t s) {
    uvs_noise_params np = np_in;
    const int variant = noise_variant(np_in);
    np.type &= 0xff;                                               // the kernels see the plain noise type
    uvs::NoiseArgs A{np, T, states, zig, out};
    const long long lanes = T * (np.m / 2);
    const dim3 g((unsigned)((lanes + 63) / 64));
    switch (np.type) {                                             // one instantiation per noise type (noise.py:7-12)
        case UVS_NOISE_WHITE: hipLaunchKernelGGL(uvs::noise_kernel<UVS_NOISE_WHITE>, g, dim3(64), 0, s, A); break;
        case UVS_NOISE_GAUSSIAN_MIXTURE: hipLaunchKernelGGL(uvs::noise_kernel<UVS_NOISE_GAUSSIAN_MIXTURE>, g, dim3(64), 0, s, A); break;
        case UVS_NOISE_GAUSSIAN_BIMODAL: hipLaunchKernelGGL(uvs::noise_kernel<UVS_NOISE_GAUSSIAN_BIMODAL>, g, dim3(64), 0, s, A); break;
        case UVS_NOISE_ALPHA_STABLE:
            if (variant == 2)
                hipLaunchKernelGGL(uvs::noise_kernel<uvs::kNoiseStableAsWritten>, g, dim3(64), 0, s, A);
            else if (variant == 1)
                hipLaunchKernelGGL(uvs::noise_kernel<uvs::kNoiseStableSymmetric>, g, dim3(64), 0, s, A);
            else
                hipLaunchKernelGGL(uvs::noise_kernel<UVS_NOISE_ALPHA_STABLE>, g, dim3(64), 0, s, A);
            break;
        default: hipLaunchKernelGGL(uvs::noise_kernel<UVS_NOISE_UNIFORM>, g, dim3(64), 0, s, A); break;
    }
}

void uvs_launch::noise_streams(const uvs_noise_params &np_in, long long S, const unsigned long long *states, const double *zig, uvs::View out, hipStream_t s) {
    uvs_noise_params np = np_in;
    const int variant = noise_variant(np_in);
    np.type &= 0xff;
    uvs::NoiseArgs A{np, S, states, zig, out};
    // draws per sample of the type's generator (noise.py:179-205): uniform and Cauchy 1, the Chambers-Mallows-Stuck transform 2 (V, W); the ziggurat
    // normal (white noise, alpha = 2, the Levy case) consumes a data-dependent number and keeps one lane per stream
    int draws = 0;
    if (np.type == UVS_NOISE_UNIFORM) draws = 1;
    else if (np.type == UVS_NOISE_ALPHA_STABLE) {
        if (np.alpha == 2.0 || (np.alpha == 0.5 && (np.beta == 1.0 || np.beta == -1.0))) draws = 0;
        else draws = (np.alpha == 1.0 && np.beta == 0.0) ? 1 : 2;
    }
    A.draws_per_sample = draws;
    A.chunks = 1;
    if (draws > 0 && S > 0) {                                     // aim at ~4 wavefronts per SIMD, at least 8 steps per chunk
        long long c = (4096LL * 64 + S - 1) / S;
        if (c > 8) c = 8;
        if (c > np.steps / 8) c = np.steps / 8;
        A.chunks = c < 1 ? 1 : (int)c;
    }
    const dim3 g((unsigned)((S * A.chunks + 63) / 64));
    switch (np.type) {
        case UVS_NOISE_WHITE: hipLaunchKernelGGL(uvs::noise_streams_kernel<UVS_NOISE_WHITE>, g, dim3(64), 0, s, A); break;
        case UVS_NOISE_ALPHA_STABLE:
            if (variant == 2)
                hipLaunchKernelGGL(uvs::noise_streams_kernel<uvs::kNoiseStableAsWritten>, g, dim3(64), 0, s, A);
            else if (variant == 1)
                hipLaunchKernelGGL(uvs::noise_streams_kernel<uvs::kNoiseStableSymmetric>, g, dim3(64), 0, s, A);
            else
                hipLaunchKernelGGL(uvs::noise_streams_kernel<UVS_NOISE_ALPHA_STABLE>, g, dim3(64), 0, s, A);
            break;
        default: hipLaunchKernelGGL(uvs::noise_streams_kernel<UVS_NOISE_UNIFORM>, g, dim3(64), 0, s, A); break;
    }
}

int uvs_launch::noise_variant(const uvs_noise_params &np) {
    if ((np.type & 0xff) != UVS_NOISE_ALPHA_STABLE) return 0;
    // as written: only the general Chambers-Mallows-Stuck branches differ (alpha = 2, Cauchy, Levy and alpha = 1 with skew already call the library)
    const bool general = np.alpha != 2.0 && np.alpha != 1.0 && !(np.alpha == 0.5 && (np.beta == 1.0 || np.beta == -1.0));
    if ((np.type & UVS_NOISE_OPT_AS_WRITTEN) && general) return 2;
    return uvs::stable_symmetric_fast(np) ? 1 : 0;
}

void uvs_launch::pcg64_seed(long long n, const unsigned long long *seeds, unsigned long long *states, hipStream_t s) {
    hipLaunchKernelGGL(uvs::pcg64_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, seeds, states);
}

// Tuned replay kernels (rmckf_replay_tuned.hpp): estimator, "X stream wanted" and "control law wanted" are compile-time.
#include "launchers.hpp"
#include "rmckf_replay_tuned.hpp"

#ifndef UVS_REPLAY_PV
#define UVS_REPLAY_PV 2
#endif

namespace {
template <int M, int N, int METHOD>
void tuned2(bool xo, bool cmd, dim3 g, hipStream_t s, const uvs::ReplayArgs &A) {
    constexpr int PV = UVS_REPLAY_PV;
    if (xo && cmd) hipLaunchKernelGGL((uvs::replay_tuned_kernel<M, N, METHOD, PV, true, true>), g, dim3(64), 0, s, A);
    else if (xo) hipLaunchKernelGGL((uvs::replay_tuned_kernel<M, N, METHOD, PV, true, false>), g, dim3(64), 0, s, A);
    else if (cmd) hipLaunchKernelGGL((uvs::replay_tuned_kernel<M, N, METHOD, PV, false, true>), g, dim3(64), 0, s, A);
    else hipLaunchKernelGGL((uvs::replay_tuned_kernel<M, N, METHOD, PV, false, false>), g, dim3(64), 0, s, A);
}
template <int M, int N, int METHOD>
void rows2(bool xo, bool eo, dim3 g, hipStream_t s, const uvs::ReplayArgs &A) {
    if (xo && eo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, true, true>), g, dim3(64), 0, s, A);
    else if (xo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, true, false>), g, dim3(64), 0, s, A);
    else if (eo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, false, true>), g, dim3(64), 0, s, A);
    else hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, false, false>), g, dim3(64), 0, s, A);
}
}  // namespace

bool uvs_launch::replay_tuned(int m, int n, int method, bool xo, bool cmd, int64_t T, hipStream_t s, const uvs::ReplayArgs &A) {
#define XR(M, N) \
    if (m == M && n == N) { \
        if (method == UVS_METHOD_GMCKF) tuned2<M, N, UVS_METHOD_GMCKF>(xo, cmd, grid_for(T, 2), s, A); \
        else if (method == UVS_METHOD_MCKF) tuned2<M, N, UVS_METHOD_MCKF>(xo, cmd, grid_for(T, 2), s, A); \
        else if (method == UVS_METHOD_IMCCKF) tuned2<M, N, UVS_METHOD_IMCCKF>(xo, cmd, grid_for(T, 2), s, A); \
        else tuned2<M, N, UVS_METHOD_KF>(xo, cmd, grid_for(T, 2), s, A); \
        return true; \
    }
    UVS_TUNED_REPLAY_SHAPES(XR)
#undef XR
    return false;
}

// Estimator-only replay, four lanes per filter, state in registers, two wavefronts per SIMD: (8,6) only.
bool uvs_launch::replay_rows(int m, int n, int method, bool xo, bool eo, int64_t T, hipStream_t s, const uvs::ReplayArgs &A) {
    if (m != 8 || n != 6) return false;
    if (method == UVS_METHOD_GMCKF) rows2<8, 6, UVS_METHOD_GMCKF>(xo, eo, grid_for(T, 4), s, A);
    else if (method == UVS_METHOD_MCKF) rows2<8, 6, UVS_METHOD_MCKF>(xo, eo, grid_for(T, 4), s, A);
    else if (method == UVS_METHOD_IMCCKF) rows2<8, 6, UVS_METHOD_IMCCKF>(xo, eo, grid_for(T, 4), s, A);
    else rows2<8, 6, UVS_METHOD_KF>(xo, eo, grid_for(T, 4), s, A);
    return true;
}

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
void rows_bywave(bool xo, bool eo, dim3 g, hipStream_t s, const uvs::ReplayArgs &A) {
    if (xo && eo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, true, true, true>), g, dim3(256), 0, s, A);
    else if (xo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, true, false, true>), g, dim3(256), 0, s, A);
    else if (eo) hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, false, true, true>), g, dim3(256), 0, s, A);
    else hipLaunchKernelGGL((uvs::replay_rows_kernel<M, N, 4, METHOD, false, false, true>), g, dim3(256), 0, s, A);
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
// Three mappings of the same arithmetic.  Record streams (X and err both [step][trial][component], whole wavefronts of 16 trials, 16-byte
// aligned): lane groups + LDS transposition, 1 KB stores.  Otherwise KF / RMCKF at lanes_per_filter = 0: the four row groups of a filter in
// the four wavefronts of a workgroup of 64 trials (512-byte stores in the trial-fastest layout).  Otherwise four lane groups of one wavefront.
// Estimator in four row-group wavefronts + control law in two more (KF / RMCKF, X, err and the commanded dq all wanted).
bool uvs_launch::replay_rows_cmd(int m, int n, int method, int64_t T, hipStream_t s, const uvs::ReplayArgs &A) {
    if (m != 8 || n != 6 || !(method == UVS_METHOD_GMCKF || method == UVS_METHOD_KF)) return false;
    const dim3 g((unsigned)((T + 63) / 64));
    if (method == UVS_METHOD_GMCKF) hipLaunchKernelGGL((uvs::replay_rows_kernel<8, 6, 4, UVS_METHOD_GMCKF, true, true, true, false, 2>), g, dim3(384), 0, s, A);
    else hipLaunchKernelGGL((uvs::replay_rows_kernel<8, 6, 4, UVS_METHOD_KF, true, true, true, false, 2>), g, dim3(384), 0, s, A);
    return true;
}

bool uvs_launch::replay_rows(int m, int n, int method, bool bywave, bool xo, bool eo, int64_t T, hipStream_t s, const uvs::ReplayArgs &A) {
    if (m != 8 || n != 6) return false;
    const bool rec = xo && eo && T % 16 == 0 && A.x_out.sc == 1 && A.x_out.st == 48 && A.err_out.sc == 1 && A.err_out.st == 8 &&
                     A.x_out.sk % 2 == 0 && A.err_out.sk % 2 == 0 && ((uintptr_t)A.x_out.p | (uintptr_t)A.err_out.p) % 16 == 0;
    if (rec) {
        const dim3 g = grid_for(T, 4);
        if (method == UVS_METHOD_GMCKF) hipLaunchKernelGGL((uvs::replay_rows_kernel<8, 6, 4, UVS_METHOD_GMCKF, true, true, false, true>), g, dim3(64), 0, s, A);
        else if (method == UVS_METHOD_MCKF) hipLaunchKernelGGL((uvs::replay_rows_kernel<8, 6, 4, UVS_METHOD_MCKF, true, true, false, true>), g, dim3(64), 0, s, A);
        else if (method == UVS_METHOD_IMCCKF) hipLaunchKernelGGL((uvs::replay_rows_kernel<8, 6, 4, UVS_METHOD_IMCCKF, true, true, false, true>), g, dim3(64), 0, s, A);
        else hipLaunchKernelGGL((uvs::replay_rows_kernel<8, 6, 4, UVS_METHOD_KF, true, true, false, true>), g, dim3(64), 0, s, A);
        return true;
    }
    if (bywave && (method == UVS_METHOD_GMCKF || method == UVS_METHOD_KF)) {
        const dim3 g((unsigned)((T + 63) / 64));
        if (method == UVS_METHOD_GMCKF) rows_bywave<8, 6, UVS_METHOD_GMCKF>(xo, eo, g, s, A);
        else rows_bywave<8, 6, UVS_METHOD_KF>(xo, eo, g, s, A);
        return true;
    }
    if (method == UVS_METHOD_GMCKF) rows2<8, 6, UVS_METHOD_GMCKF>(xo, eo, grid_for(T, 4), s, A);
    else if (method == UVS_METHOD_MCKF) rows2<8, 6, UVS_METHOD_MCKF>(xo, eo, grid_for(T, 4), s, A);
    else if (method == UVS_METHOD_IMCCKF) rows2<8, 6, UVS_METHOD_IMCCKF>(xo, eo, grid_for(T, 4), s, A);
    else rows2<8, 6, UVS_METHOD_KF>(xo, eo, grid_for(T, 4), s, A);
    return true;
}

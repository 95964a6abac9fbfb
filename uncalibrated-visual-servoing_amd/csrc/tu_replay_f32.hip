// Single-precision estimator-only replay (rmckf_replay_f32.hpp): estimator and "stream wanted" flags are compile-time.
#include "launchers.hpp"
#include "rmckf_replay_f32.hpp"

namespace {
template <int METHOD>
void launch_f32(bool xo, bool eo, dim3 g, hipStream_t s, const uvs::ReplayArgs32 &A) {
    if (xo && eo) hipLaunchKernelGGL((uvs::replay_f32_kernel<METHOD, true, true>), g, dim3(64), 0, s, A);
    else if (xo) hipLaunchKernelGGL((uvs::replay_f32_kernel<METHOD, true, false>), g, dim3(64), 0, s, A);
    else if (eo) hipLaunchKernelGGL((uvs::replay_f32_kernel<METHOD, false, true>), g, dim3(64), 0, s, A);
    else hipLaunchKernelGGL((uvs::replay_f32_kernel<METHOD, false, false>), g, dim3(64), 0, s, A);
}
}  // namespace

bool uvs_launch::replay_f32(int m, int n, int method, int64_t T, hipStream_t s, const uvs::ReplayArgs32 &A) {
    if (m != 8 || n != 6) return false;
    const bool xo = A.x_out.p != nullptr, eo = A.err_out.p != nullptr;
    const dim3 g = grid_for(T, 2);
    if (method == UVS_METHOD_GMCKF) launch_f32<UVS_METHOD_GMCKF>(xo, eo, g, s, A);
    else if (method == UVS_METHOD_IMCCKF) launch_f32<UVS_METHOD_IMCCKF>(xo, eo, g, s, A);
    else if (method == UVS_METHOD_KF) launch_f32<UVS_METHOD_KF>(xo, eo, g, s, A);
    else return false;
    return true;
}

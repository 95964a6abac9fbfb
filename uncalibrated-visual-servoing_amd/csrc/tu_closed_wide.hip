// Tuned closed-loop kernel of the wide stress shape (rmckf_wide.hpp): (32,7), 8 lanes per filter, linear plant, estimator in
// {KF, IMCCKF, GMCKF} -- and the same kernel on the DH / pinhole plant at (8,6), one row per lane: lanes_per_filter = 8.
#include "launchers.hpp"
#include "rmckf_wide.hpp"

namespace {
template <int M, int N, int LL, int METHOD>
void launch_wide(bool xo, dim3 g, hipStream_t s, const uvs::ClosedArgs &A) {
    // per-trial records ([step][trial][component]) and a whole number of wavefronts: X leaves through the LDS transposition as 1 KB stores
    // (the record path stores double2: the base must be 16-byte aligned and the step stride even, or the strided variant takes over)
    const bool rec = xo && A.x_out.sc == 1 && A.x_out.st == M * N && A.T % (64 / LL) == 0 && (uintptr_t)A.x_out.p % 16 == 0 && A.x_out.sk % 2 == 0;
    if (rec) hipLaunchKernelGGL((uvs::closed_loop_wide_kernel<M, N, LL, METHOD, true, true>), g, dim3(64), 0, s, A);
    else if (xo) hipLaunchKernelGGL((uvs::closed_loop_wide_kernel<M, N, LL, METHOD, true, false>), g, dim3(64), 0, s, A);
    else hipLaunchKernelGGL((uvs::closed_loop_wide_kernel<M, N, LL, METHOD, false, false>), g, dim3(64), 0, s, A);
}
template <int METHOD>
void launch_latency(bool xo, dim3 g, hipStream_t s, const uvs::ClosedArgs &A) {
    if (xo) hipLaunchKernelGGL((uvs::closed_loop_wide_kernel<8, 6, 8, METHOD, true, false, UVS_PLANT_DH_PINHOLE>), g, dim3(64), 0, s, A);
    else hipLaunchKernelGGL((uvs::closed_loop_wide_kernel<8, 6, 8, METHOD, false, false, UVS_PLANT_DH_PINHOLE>), g, dim3(64), 0, s, A);
}
}  // namespace

bool uvs_launch::closed_wide(int m, int n, int L, int method, bool linear, bool xo, int64_t T, hipStream_t s, const uvs::ClosedArgs &A) {
    if (m == 8 && n == 6 && L == 8 && !linear) {                  // (explicit lanes_per_filter = 8 only: measured, not faster than four lanes)
        const dim3 g8 = grid_for(T, L);
        if (method == UVS_METHOD_GMCKF) launch_latency<UVS_METHOD_GMCKF>(xo, g8, s, A);
        else if (method == UVS_METHOD_IMCCKF) launch_latency<UVS_METHOD_IMCCKF>(xo, g8, s, A);
        else if (method == UVS_METHOD_KF) launch_latency<UVS_METHOD_KF>(xo, g8, s, A);
        else return false;
        return true;
    }
    if (m != 32 || n != 7 || (L != 8 && L != 16) || !linear || A.fp.initial_guess) return false;
    const dim3 g = grid_for(T, L);
    if (L == 8) {
        if (method == UVS_METHOD_GMCKF) launch_wide<32, 7, 8, UVS_METHOD_GMCKF>(xo, g, s, A);
        else if (method == UVS_METHOD_IMCCKF) launch_wide<32, 7, 8, UVS_METHOD_IMCCKF>(xo, g, s, A);
        else if (method == UVS_METHOD_KF) launch_wide<32, 7, 8, UVS_METHOD_KF>(xo, g, s, A);
        else return false;
    } else {
        if (method == UVS_METHOD_GMCKF) launch_wide<32, 7, 16, UVS_METHOD_GMCKF>(xo, g, s, A);
        else if (method == UVS_METHOD_IMCCKF) launch_wide<32, 7, 16, UVS_METHOD_IMCCKF>(xo, g, s, A);
        else if (method == UVS_METHOD_KF) launch_wide<32, 7, 16, UVS_METHOD_KF>(xo, g, s, A);
        else return false;
    }
    return true;
}

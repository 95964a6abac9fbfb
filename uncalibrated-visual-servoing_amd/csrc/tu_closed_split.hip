// Role-split closed-loop kernel (rmckf_split.hpp): (8,6), DH/pinhole plant, estimator in {KF, IMCCKF, GMCKF}; lanes_per_filter = 5
// ("4 estimator lanes + 1 control lane per trial").
#include "launchers.hpp"
#include "rmckf_split.hpp"

namespace {
template <int M, int N, int METHOD>
void launch_split(bool xo, dim3 g, hipStream_t s, const uvs::ClosedArgs &A) {
    if (xo) hipLaunchKernelGGL((uvs::closed_loop_split_kernel<M, N, METHOD, true>), g, dim3(256), 0, s, A);
    else hipLaunchKernelGGL((uvs::closed_loop_split_kernel<M, N, METHOD, false>), g, dim3(256), 0, s, A);
}
}  // namespace

bool uvs_launch::closed_split(int m, int n, int method, bool linear, bool xo, int64_t T, hipStream_t s, const uvs::ClosedArgs &A) {
    if (m != 8 || n != 6 || linear) return false;
    // the kernel addresses a trial inside a stream row by a 32-bit byte offset
    for (const uvs::View *v : {&A.noise, &A.x_out, &A.err_out, &A.q_out, &A.f_out, &A.dq_out})
        if (v->p && (v->st < 0 || (unsigned long long)(T - 1) * (unsigned long long)v->st * 8ull >= (1ull << 32))) return false;
    const dim3 g((unsigned)((T + 63) / 64));
    if (method == UVS_METHOD_GMCKF) launch_split<8, 6, UVS_METHOD_GMCKF>(xo, g, s, A);
    else if (method == UVS_METHOD_IMCCKF) launch_split<8, 6, UVS_METHOD_IMCCKF>(xo, g, s, A);
    else if (method == UVS_METHOD_KF) launch_split<8, 6, UVS_METHOD_KF>(xo, g, s, A);
    else return false;
    return true;
}

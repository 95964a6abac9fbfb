// Host-side launch functions, one translation unit per kernel family so that the library builds in parallel (make -j):
// the C ABI (uvs_rmckf.hip) validates arguments and asks each family in turn; a launcher returns false when it has no
// instantiation for the request.  The shape tables live here.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "rmckf_device.hpp"

namespace uvs { struct ReplayArgs32; }

// (m, n, lanes-per-filter) instantiations of the generic templates; the first listed L of a shape is its default.
#ifdef UVS_QUICK                      // experiment builds (make quick): the headline shape only, compiles in seconds
#define UVS_SHAPES_A(X) X(8, 6, 2) X(8, 6, 4)
#define UVS_SHAPES_B(X)
#define UVS_TUNED_SHAPES_A(X) X(8, 6, 2)
#ifdef UVS_QUICK_L4                   // ... plus the four-lane tuned kernel (latency experiments)
#define UVS_TUNED_SHAPES_B(X) X(8, 6, 4)
#else
#define UVS_TUNED_SHAPES_B(X)
#endif
#else
#define UVS_SHAPES_A(X) X(8, 6, 2) X(8, 6, 1) X(8, 6, 4) X(8, 6, 8) X(2, 6, 1)
#define UVS_SHAPES_B(X) X(6, 6, 2) X(6, 6, 1) X(32, 7, 16) X(32, 7, 32) X(32, 7, 8)
#define UVS_TUNED_SHAPES_A(X) X(8, 6, 2) X(6, 6, 2)
#define UVS_TUNED_SHAPES_B(X) X(8, 6, 1) X(8, 6, 4)
#endif
#define UVS_SHAPES(X) UVS_SHAPES_A(X) UVS_SHAPES_B(X)
#if !defined(UVS_QUICK) || defined(UVS_QUICK_L4)
#define UVS_HAVE_EMU2 1               // the four-lane kernels with the two-lane kernel's bits are in this build (tu_closed_tuned_b.hip)
#endif
// one careful (numpy-pinv) instantiation of the generic closed-loop / replay kernels per shape: the second pass over suspect trials
#ifdef UVS_QUICK
#define UVS_CAREFUL_SHAPES_A(X) X(8, 6, 4)
#define UVS_CAREFUL_SHAPES_B(X)
#else
#define UVS_CAREFUL_SHAPES_A(X) X(8, 6, 4) X(2, 6, 1)
#define UVS_CAREFUL_SHAPES_B(X) X(6, 6, 2) X(32, 7, 16)
#endif
#define UVS_TUNED_REPLAY_SHAPES(X) X(8, 6) X(6, 6)

namespace uvs_launch {

inline dim3 grid_for(int64_t T, int L) { return dim3((unsigned)((T * L + 63) / 64)); }

// tuned closed loop (rmckf_tuned.hpp): method in {KF, IMCCKF, GMCKF}
bool closed_tuned_a(int m, int n, int L, int method, bool linear, bool xo, int64_t T, hipStream_t s, const uvs::ClosedArgs &A);
bool closed_tuned_b(int m, int n, int L, int method, bool linear, bool xo, int64_t T, hipStream_t s, const uvs::ClosedArgs &A);
// four lanes per filter with the two-lane kernel's bits (EMU2): (8,6), DH plant, KF / IMCC-KF / RMCKF -- the automatic choice for batches that do not fill the chip
bool closed_tuned_emu2(int m, int n, int method, bool linear, bool xo, int64_t T, hipStream_t s, const uvs::ClosedArgs &A);
// tuned wide-shape closed loop (rmckf_wide.hpp): (32,7), lanes_per_filter = 8 (the default of that shape for the closed loop)
bool closed_wide(int m, int n, int L, int method, bool linear, bool xo, int64_t T, hipStream_t s, const uvs::ClosedArgs &A);
// generic templates (rmckf_generic.hpp)
bool closed_generic_a(int m, int n, int L, int method, int64_t T, hipStream_t s, const uvs::ClosedArgs &A);
bool closed_generic_b(int m, int n, int L, int method, int64_t T, hipStream_t s, const uvs::ClosedArgs &A);
bool replay_generic_a(int m, int n, int L, int method, int64_t T, hipStream_t s, const uvs::ReplayArgs &A);
bool replay_generic_b(int m, int n, int L, int method, int64_t T, hipStream_t s, const uvs::ReplayArgs &A);
bool step_generic(int m, int n, int L, int64_t T, hipStream_t s, const uvs::StepArgs &A);
// careful second pass (rmckf_generic.hpp, CAREFUL = true): re-runs the trials whose status the first pass left at UVS_STATUS_SUSPECT
bool closed_careful_a(int m, int n, int64_t T, hipStream_t s, const uvs::ClosedArgs &A);
bool closed_careful_b(int m, int n, int64_t T, hipStream_t s, const uvs::ClosedArgs &A);
bool replay_careful_a(int m, int n, int64_t T, hipStream_t s, const uvs::ReplayArgs &A);
bool replay_careful_b(int m, int n, int64_t T, hipStream_t s, const uvs::ReplayArgs &A);
// tuned replay (rmckf_replay_tuned.hpp): two lanes per filter with the control law, four lanes per filter for the estimator alone
bool replay_tuned(int m, int n, int method, bool xo, bool cmd, int64_t T, hipStream_t s, const uvs::ReplayArgs &A);
bool replay_rows_cmd(int m, int n, int method, int64_t T, hipStream_t s, const uvs::ReplayArgs &A);
bool replay_rows(int m, int n, int method, bool bywave, bool xo, bool eo, int64_t T, hipStream_t s, const uvs::ReplayArgs &A);
// single-precision estimator-only replay (rmckf_replay_f32.hpp): (8,6), KF / IMCC-KF / RMCKF
bool replay_f32(int m, int n, int method, int64_t T, hipStream_t s, const uvs::ReplayArgs32 &A);
// everything else
void stats(long long T, int K, int m, uvs::View err, const double *t, const int *k_done, double *stats, hipStream_t s);
void debug_math(int which, long long n, const double *x, double *y, hipStream_t s);
void noise(const uvs_noise_params &np, long long T, const unsigned long long *states, const double *zig, uvs::View out, hipStream_t s);
void noise_streams(const uvs_noise_params &np, long long S, const unsigned long long *states, const double *zig, uvs::View out, hipStream_t s);   // one generator per stream, no hold
int noise_variant(const uvs_noise_params &np);                       // 0 = kernel of np.type, 1 = beta = 0 alpha-stable specialisation
void pcg64_seed(long long n, const unsigned long long *seeds, unsigned long long *states, hipStream_t s);
// p[0..n) = v as a KERNEL: inside a captured graph a memset node in front of the closed-loop kernel was seen to land late on every other replay
// (ROCm 7.2, tests/test_gpu_graph.py), a kernel node keeps its place in the stream's order
void fill_i32(int *p, int v, long long n, hipStream_t s);

}  // namespace uvs_launch

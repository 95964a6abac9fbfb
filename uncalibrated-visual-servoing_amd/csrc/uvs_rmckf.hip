// libuvs_rmckf.so -- C ABI (include/uvs_rmckf.h) of the batched RMCKF estimator, gfx950 only: argument checks and dispatch.
//
// Kernels (all fp64, wave64, 64-thread workgroups so that each wavefront is scheduled independently) live in headers and are
// instantiated by one translation unit per family (launchers.hpp) so that the library builds in parallel:
//   rmckf_tuned.hpp          closed_loop_tuned_kernel: whole servo trial per filter, headline shapes (tu_closed_tuned_{a,b}.hip)
//   rmckf_replay_tuned.hpp   replay_tuned_kernel / replay_rows_kernel: estimator (+ control law) over recorded streams (tu_replay_tuned.hip)
//   rmckf_generic.hpp        closed_loop_kernel / replay_kernel / step_kernel: any shape, every estimator incl. MCKF; stats_kernel
//                            (tu_generic_{a,b}.hip, tu_misc.hip)
//   noise_kernels.hpp        noise_kernel, pcg64_seed_kernel (tu_misc.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "uvs_rmckf.h"
#include "launchers.hpp"
#include "rmckf_replay_f32.hpp"

#ifndef UVS_MCKF_TAPER_PCT               // length of the last segment of an MCKF trial in % of the first (linear in between): 100 / 50 / 25 / 10 % measured
#define UVS_MCKF_TAPER_PCT 10            // 4.32 / 4.12 / 4.07 / 4.04 ms at 8 segments (unsegmented 4.43), 4.96 / 4.91 / 4.73 / 4.64 on alpha = 1.0 (5.51)
#endif

using namespace uvs_launch;

// ================================================================================================ C ABI
namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    std::snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
        return UVS_ERR_HIP;
    }
    return UVS_OK;
}

int default_lanes(int m, int n, int method) {
    // (MCKF: the tuned two-lane kernels run the first fixed-point pass and leave trials that need more to the careful second pass,
    // whose generic template carries the Cholesky factors of the blocks -- four lanes per filter there, at two they go to scratch)
    (void)method;
#define X(M, N, L) if (m == M && n == N) return L;
    UVS_SHAPES(X)
#undef X
    return 0;
}

int check_params(const uvs_filter_params *fp, int64_t T, int *lanes) {
    if (!fp) return fail(UVS_ERR_ARG, "%s", "filter params are NULL");
    if (T <= 0) return fail(UVS_ERR_ARG, "%s", "T must be positive");
    if (fp->steps < 0 || fp->k_max <= 0) return fail(UVS_ERR_ARG, "%s", "steps must be >= 0 and k_max > 0");
    if (fp->method != UVS_METHOD_KF && fp->method != UVS_METHOD_MCKF && fp->method != UVS_METHOD_IMCCKF && fp->method != UVS_METHOD_GMCKF)
        return fail(UVS_ERR_METHOD, "%s", "method must be KF, MCKF, IMCCKF or GMCKF");
    if (fp->method == UVS_METHOD_MCKF && fp->fpi_epoch_max < 1) return fail(UVS_ERR_ARG, "%s", "MCKF needs fpi_epoch_max >= 1");
    const int L = fp->lanes_per_filter < 0 ? -fp->lanes_per_filter : (fp->lanes_per_filter ? fp->lanes_per_filter : default_lanes(fp->m, fp->n, fp->method));
    if (L == 0) return fail(UVS_ERR_SHAPE, "%s", "(m, n) is not instantiated in libuvs_rmckf");
    *lanes = L;
    return UVS_OK;
}

// Four lanes per filter for a closed-loop batch of the (8,6) shape that four-lane wavefronts still run in one round (1024 SIMDs, one wavefront
// each, 16 trials per wavefront): half the trials per wavefront, a shorter step, 20-28 % less time per launch.  By default
// -- lanes_per_filter == 0, KF / IMCC-KF / RMCKF on the DH plant -- the EMU2 kernels, which reproduce the two-lane arithmetic bit for bit, so
// the choice is invisible in the results; with UVS_OPT_LATENCY the plain four-lane kernels (3-7 % faster still, last-bit differences).  MCKF
// (round 5) has the EMU2 kernel too -- every fixed-point pass in-kernel -- but no plain four-lane one.  Returns 0 = no change, 4 = plain four lanes,
// -4 = four lanes with the two-lane bits.
int small_batch_lanes(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T) {
    if (fp->lanes_per_filter != 0 || fp->m != 8 || fp->n != 6 || (T * 4 + 63) / 64 > 1024) return 0;
    if (((fp->reserved >> 8) & 0xff) > 1) return 0;                   // a forced segment count (testing / measurements) asks for the segmented two-lane kernel
    // (eight lanes per filter -- one row per lane on the wide kernel's DH instantiation, lanes_per_filter = 8 -- were built and measured for this
    // role: 8 192 trials 0.92 ms against 0.88 ms on four lanes -- the plant replicated on eight lanes gives back what one row per lane saves
    // (1 075 against 1 109 VALU instructions per wavefront-step): DESIGN.md section 6)
    if (fp->reserved & UVS_OPT_LATENCY) return fp->method == UVS_METHOD_MCKF ? 0 : 4;   // (the plain four-lane kernels run only MCKF's first pass: no latency mapping for it)
#ifdef UVS_HAVE_EMU2
    if (plant->kind == UVS_PLANT_DH_PINHOLE && !(fp->reserved & UVS_OPT_STRICT_PINV)) return -4;
#endif
    return 0;
}

}  // namespace

extern "C" {

#ifndef UVS_SRC_HASH
#define UVS_SRC_HASH "unknown"
#endif
const char *uvs_version(void) { return "uvs_rmckf 0.6.0 (gfx950, fp64) src:" UVS_SRC_HASH; }
const char *uvs_last_error(void) { return g_err; }

int uvs_supported_lanes(int32_t m, int32_t n, int32_t *lanes, int32_t cap) {
    int cnt = 0;
#define X(M, N, L) if (m == M && n == N) { if (lanes && cnt < cap) lanes[cnt] = L; ++cnt; }
    UVS_SHAPES(X)
#undef X
    return cnt;
}

// Segmented trials (tuned two-lane MCKF kernel).  How many segments a launch of T trials is cut into: MCKF wavefronts differ in length (a
// trial whose fixed-point iteration keeps iterating costs its whole wavefront the branch), so the last round of a launch of whole trials
// leaves SIMDs idle for up to a third of a trial.  Bits 8-15 of fp->reserved override (1 = never, n = n segments).
namespace {
constexpr int64_t kSimdSlots = 1024;
int segments_for(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T) {
    if (!fp || !plant || T <= 0 || fp->lanes_per_filter < 0 || plant->kind != UVS_PLANT_DH_PINHOLE) return 1;
    const bool mckf = fp->method == UVS_METHOD_MCKF, rmckf = fp->method == UVS_METHOD_GMCKF && fp->m == 8 && fp->n == 6;
    if (!mckf && !rmckf) return 1;
    if (fp->reserved & UVS_OPT_STRICT_PINV) return 1;                // (every trial goes to the careful kernels: nothing to cut)
    const int L = fp->lanes_per_filter ? fp->lanes_per_filter : default_lanes(fp->m, fp->n, fp->method);
    bool tuned2 = false;
#define X(M, N, LL) if (fp->m == M && fp->n == N && L == LL && LL == 2) tuned2 = true;
    UVS_TUNED_SHAPES_A(X) UVS_TUNED_SHAPES_B(X)
#undef X
    if (!tuned2) return 1;
    const int forced = (fp->reserved >> 8) & 0xff;
    int n = forced;
    const int64_t chunks = (T * L + 63) / 64;
    if (!n && mckf) {
        // measured on MI355X (32 trials per wavefront, one wavefront per SIMD): one round or less -- nothing to balance;
        // up to three rounds -- 8 segments (49 152 trials 4.02 -> 3.15 ms, 65 536: 4.43 -> 4.04, 98 304: 6.46 -> 5.94); beyond -- 4
        // (131 072: 8.24 -> 7.66, 262 144: 15.6 -> 15.0), where 8 hand-overs per chunk cost more than the shorter tail returns
        n = chunks <= kSimdSlots ? 1 : (chunks <= 3 * kSimdSlots ? 8 : 4);
    }
    if (!n && rmckf) {
        // RMCKF wavefronts all take the same time, so only a launch that is NOT a whole number of rounds has something to balance: 1.5 rounds take
        // two rounds' time as whole trials (49 152 trials 2.97 -> 2.65 ms, 81 920: 4.49 -> 4.11 in four segments); whole rounds (the BASELINE configs)
        // and launches beyond six rounds keep whole trials and the instantiation without the hand-over code
        const int64_t over = chunks % kSimdSlots;
        n = (chunks > kSimdSlots && chunks < 6 * kSimdSlots && over >= kSimdSlots / 10 && over <= kSimdSlots - kSimdSlots / 10) ? 4 : 1;
    }
    if (small_batch_lanes(fp, plant, T)) n = 1;                      // (small batches run on four lanes per filter: no segmented kernel there, none needed)
    if (n > 16) n = 16;
    if (n > 1 && fp->steps < 8 * n) n = 1;                        // nothing to cut in a short trial
    return n < 1 ? 1 : n;
}
size_t seg_flag_bytes(int64_t chunks) { return (size_t)(((chunks + 1) * sizeof(int) + 255) / 256) * 256; }   // one counter per chunk + the fallback count
}  // namespace

int uvs_rmckf_closed_loop_segments(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T) { return segments_for(fp, plant, T); }

int uvs_rmckf_closed_loop_lanes(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T) {
    int L = 0;
    if (!fp || !plant || T <= 0 || check_params(fp, T, &L) != UVS_OK) return 0;
    if (fp->lanes_per_filter == 0 && fp->m == 32 && fp->n == 7 && plant->kind == UVS_PLANT_LINEAR && !fp->initial_guess && fp->method != UVS_METHOD_MCKF) return 8;
    return small_batch_lanes(fp, plant, T) ? 4 : L;
}

size_t uvs_rmckf_closed_loop_workspace_bytes(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T) {
    const int n = segments_for(fp, plant, T);
    if (n <= 1) return 0;
    const int64_t chunks = (T * 2 + 63) / 64;
    return seg_flag_bytes(chunks) + (size_t)chunks * uvs::seg_state_doubles(fp->m, fp->n, 2) * 64 * sizeof(double);
}

size_t uvs_rmckf_closed_loop_fallback_offset(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T) {
    if (segments_for(fp, plant, T) <= 1) return 0;
    return (size_t)((T * 2 + 63) / 64) * sizeof(int);
}

int uvs_rmckf_closed_loop_f64(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T, uvs_view q_start, uvs_view noise,
                              uvs_view x0, uvs_view x_out, uvs_view err_out, uvs_view q_out, uvs_view f_out, uvs_view dq_out,
                              double *stats, int32_t *status, int32_t *k_done, uvs_view x_final, uvs_view p_final, void *stream) {
    return uvs_rmckf_closed_loop_ws_f64(fp, plant, T, q_start, noise, x0, x_out, err_out, q_out, f_out, dq_out, stats, status, k_done, x_final, p_final,
                                        nullptr, 0, stream);
}

int uvs_rmckf_closed_loop_ws_f64(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T, uvs_view q_start, uvs_view noise,
                                 uvs_view x0, uvs_view x_out, uvs_view err_out, uvs_view q_out, uvs_view f_out, uvs_view dq_out,
                                 double *stats, int32_t *status, int32_t *k_done, uvs_view x_final, uvs_view p_final,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    int L = 0;
    if (int rc = check_params(fp, T, &L)) return rc;
    if (!plant) return fail(UVS_ERR_ARG, "%s", "plant is NULL");
    if (plant->n_joints != fp->n) return fail(UVS_ERR_ARG, "%s", "plant does not match n");
    if (plant->kind == UVS_PLANT_DH_PINHOLE && plant->n_points * 2 != fp->m) return fail(UVS_ERR_ARG, "%s", "plant does not match m");
    if (plant->kind == UVS_PLANT_LINEAR) {
        if (!plant->lin_jacobian || !plant->lin_f0 || !plant->lin_q0) return fail(UVS_ERR_ARG, "%s", "linear plant arrays are NULL");
        if (fp->initial_guess) return fail(UVS_ERR_ARG, "%s", "the analytic initial guess needs the DH/pinhole plant; pass x0");
    } else if (plant->kind != UVS_PLANT_DH_PINHOLE) {
        return fail(UVS_ERR_ARG, "%s", "unknown plant kind");
    }
    if (!q_start.base) return fail(UVS_ERR_ARG, "%s", "q_start view is NULL");
    if (!status) return fail(UVS_ERR_ARG, "%s", "status is required (it also carries the suspect marks between the two passes)");
    if (!fp->initial_guess && !x0.base) return fail(UVS_ERR_ARG, "%s", "x0 view is required when initial_guess == 0");
    const int small = small_batch_lanes(fp, plant, T);
    if (small == 4) L = 4;
    uvs::ClosedArgs A;
    A.fp = *fp;
    A.plant = *plant;
    A.T = T;
    A.q_start = uvs::to_view(q_start); A.noise = uvs::to_view(noise); A.x0 = uvs::to_view(x0);
    A.x_out = uvs::to_view(x_out); A.err_out = uvs::to_view(err_out); A.q_out = uvs::to_view(q_out);
    A.f_out = uvs::to_view(f_out); A.dq_out = uvs::to_view(dq_out);
    A.x_final = uvs::to_view(x_final); A.p_final = uvs::to_view(p_final);
    A.stats = stats; A.status = status; A.k_done = k_done;
    if (workspace) {                                               // segmented trials, when the caller lent enough memory for them
        if (((uintptr_t)workspace & 7u) != 0) return fail(UVS_ERR_ARG, "%s", "workspace must be 8-byte aligned");
        const size_t need = uvs_rmckf_closed_loop_workspace_bytes(fp, plant, T);
        if (need > 0 && workspace_bytes >= need) {
            A.n_seg = segments_for(fp, plant, T);
            // Segment lengths taper linearly towards the end of the trial (the last one UVS_MCKF_TAPER_PCT % of the first): what is left
            // unbalanced at the end of the launch is one short work item per slot instead of one of average length.
            const int taper = UVS_MCKF_TAPER_PCT;
            double w[uvs::kMaxSegments], sum = 0.0;
            for (int i = 0; i < A.n_seg; ++i) { w[i] = 100.0 - (100.0 - taper) * i / (A.n_seg > 1 ? A.n_seg - 1 : 1); sum += w[i]; }
            double acc = 0.0;
            A.seg_first[0] = 0;
            for (int i = 0; i < A.n_seg; ++i) {
                acc += w[i];
                int b = (int)(fp->steps * (acc / sum) + 0.5);
                if (b <= A.seg_first[i]) b = A.seg_first[i] + 1;
                A.seg_first[i + 1] = b < fp->steps ? b : fp->steps;
            }
            A.seg_first[A.n_seg] = fp->steps;
            A.ws_flags = (int *)workspace;
            A.ws_state = (double *)((char *)workspace + seg_flag_bytes((T * 2 + 63) / 64));
        }
    }
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
    // lanes_per_filter 1 / 2 / 4 select the tuned kernel (rmckf_tuned.hpp) where it exists; a negative value forces the generic
    // template with |value| lanes (kept as an in-library cross-check of the tuned code).
    const bool tuned_ok = (fp->method == UVS_METHOD_GMCKF || fp->method == UVS_METHOD_KF || fp->method == UVS_METHOD_IMCCKF ||
                           (fp->method == UVS_METHOD_MCKF && L == 2)) && fp->lanes_per_filter >= 0;
    const bool linear = plant->kind == UVS_PLANT_LINEAR, xo = x_out.base != nullptr;
    if (fp->lanes_per_filter == 0 && fp->m == 32 && fp->n == 7 && tuned_ok && linear && !fp->initial_guess) L = 8;   // wide-shape tuned kernel
    if (fp->reserved & UVS_OPT_STRICT_PINV) {
        // numpy's pinv on every solve.  The tuned QR kernels (lanes 1 / 2 / 4 per filter; not the wide shape's normal equations) CERTIFY every
        // solve themselves in this mode -- a rigorous upper bound of the condition number from the inverse of the triangular factor, rmckf_tuned.hpp
        // lstsq_tall_tuned -- and mark what they cannot certify: one fast pass plus the careful pass for the marked trials (round 6; ~1.1 x the
        // default mode).  Everything else: mark every trial, the careful pass below is the only pass (an order of magnitude slower).
        const bool wide_takes = (fp->m == 8 && fp->n == 6 && L == 8 && !linear) || (fp->m == 32 && fp->n == 7 && (L == 8 || L == 16) && linear);
        const bool has_cert = fp->method == UVS_METHOD_GMCKF || (fp->m == 8 && fp->n == 6 && L == 2 && !linear);   // (tu_closed_tuned.inc: the CERT instantiations)
        bool certified_pass = false;
        if (tuned_ok && !wide_takes && has_cert)
            certified_pass = closed_tuned_a(fp->m, fp->n, L, fp->method, linear, xo, T, s, A) || closed_tuned_b(fp->m, fp->n, L, fp->method, linear, xo, T, s, A);
        if (!certified_pass) uvs_launch::fill_i32(status, uvs::UVS_STATUS_SUSPECT, (long long)T, s);
        launched = true;
    }
#ifdef UVS_HAVE_EMU2
    if (!launched && tuned_ok && small == -4) launched = closed_tuned_emu2(fp->m, fp->n, fp->method, linear, xo, T, s, A);
#endif
    if (!launched && tuned_ok) launched = closed_wide(fp->m, fp->n, L, fp->method, linear, xo, T, s, A);
    if (!launched && tuned_ok) launched = closed_tuned_a(fp->m, fp->n, L, fp->method, linear, xo, T, s, A) || closed_tuned_b(fp->m, fp->n, L, fp->method, linear, xo, T, s, A);
    if (!launched) launched = closed_generic_a(fp->m, fp->n, L, fp->method, T, s, A) || closed_generic_b(fp->m, fp->n, L, fp->method, T, s, A);
    if (!launched) return fail(UVS_ERR_SHAPE, "%s", "(m, n, lanes_per_filter) is not instantiated in libuvs_rmckf");
    if (int rc = check_launch("closed_loop_kernel")) return rc;
#ifdef UVS_NO_CAREFUL                  // diagnostic build: leave the marks of the first pass in `status` (how many trials does the second pass redo?)
    return UVS_OK;
#endif
    // second pass: trials in which the control law met a numerically rank-deficient Jacobian (status left at UVS_STATUS_SUSPECT) are
    // re-run with numpy's pinv semantics (experiment.py:312); wavefronts without such a trial exit at once
    if (!(closed_careful_a(fp->m, fp->n, T, s, A) || closed_careful_b(fp->m, fp->n, T, s, A)))
        return fail(UVS_ERR_SHAPE, "%s", "(m, n) has no careful closed-loop instantiation in libuvs_rmckf");
    return check_launch("closed_loop_kernel (careful pass)");
}

int uvs_rmckf_replay_f64(const uvs_filter_params *fp, int64_t T, uvs_view f, uvs_view dq, uvs_view x0, uvs_view x_out,
                         uvs_view err_out, uvs_view kappa_out, uvs_view dqcmd_out, int32_t *status, int32_t *k_done,
                         uvs_view x_final, uvs_view p_final, void *stream) {
    int L = 0;
    if (int rc = check_params(fp, T, &L)) return rc;
    if (!f.base || !dq.base || !x0.base) return fail(UVS_ERR_ARG, "%s", "f, dq and x0 views are required");
    if ((dqcmd_out.base || fp->method == UVS_METHOD_MCKF) && !status)
        return fail(UVS_ERR_ARG, "%s", "status is required when the commanded dq is requested or the estimator is MCKF (it carries the marks between the two passes)");
    uvs::ReplayArgs A;
    A.fp = *fp;
    A.T = T;
    A.f = uvs::to_view(f); A.dq = uvs::to_view(dq); A.x0 = uvs::to_view(x0);
    A.x_out = uvs::to_view(x_out); A.err_out = uvs::to_view(err_out); A.kappa_out = uvs::to_view(kappa_out);
    A.dqcmd_out = uvs::to_view(dqcmd_out); A.x_final = uvs::to_view(x_final); A.p_final = uvs::to_view(p_final);
    A.status = status; A.k_done = k_done;
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
    // two lanes per filter (the default) at (8,6): tuned kernel; a negative lanes_per_filter forces the generic template
    const bool tuned_method = fp->method == UVS_METHOD_GMCKF || fp->method == UVS_METHOD_KF || fp->method == UVS_METHOD_IMCCKF ||
                              fp->method == UVS_METHOD_MCKF;
    const bool tuned_ok = tuned_method && fp->lanes_per_filter >= 0 && L == 2;
    // without the commanded dq there is no least-squares solve and nothing couples a filter's rows: four lanes per filter, state in
    // registers, two wavefronts per SIMD (library default, or lanes_per_filter = 4)
    if (tuned_method && !dqcmd_out.base && (fp->lanes_per_filter == 0 || fp->lanes_per_filter == 4))
        launched = replay_rows(fp->m, fp->n, fp->method, fp->lanes_per_filter == 0, x_out.base != nullptr, err_out.base != nullptr, T, s, A);
    if ((fp->reserved & UVS_OPT_STRICT_PINV) && dqcmd_out.base) {   // numpy's pinv on every solve: the careful pass below is the only pass
        uvs_launch::fill_i32(status, uvs::UVS_STATUS_SUSPECT, (long long)T, s);
        launched = true;
    }
    // with the commanded dq (library default lanes, KF / RMCKF, X and err wanted too): the same estimator wavefronts + control wavefronts
    if (!launched && tuned_method && dqcmd_out.base && x_out.base && err_out.base && fp->lanes_per_filter == 0)
        launched = replay_rows_cmd(fp->m, fp->n, fp->method, T, s, A);
    if (!launched && tuned_ok) launched = replay_tuned(fp->m, fp->n, fp->method, x_out.base != nullptr, dqcmd_out.base != nullptr, T, s, A);
    if (!launched) launched = replay_generic_a(fp->m, fp->n, L, fp->method, T, s, A) || replay_generic_b(fp->m, fp->n, L, fp->method, T, s, A);
    if (!launched) return fail(UVS_ERR_SHAPE, "%s", "(m, n, lanes_per_filter) is not instantiated in libuvs_rmckf");
    if (int rc = check_launch("replay_kernel")) return rc;
    if (dqcmd_out.base || fp->method == UVS_METHOD_MCKF) {         // control law ran / first-pass MCKF rows: careful second pass over the marked trials
        if (!(replay_careful_a(fp->m, fp->n, T, s, A) || replay_careful_b(fp->m, fp->n, T, s, A)))
            return fail(UVS_ERR_SHAPE, "%s", "(m, n) has no careful replay instantiation in libuvs_rmckf");
        return check_launch("replay_kernel (careful pass)");
    }
    return UVS_OK;
}

int uvs_rmckf_replay_f32(const uvs_filter_params *fp, int64_t T, uvs_view_f32 f, uvs_view_f32 dq, uvs_view_f32 x0, uvs_view_f32 x_out,
                         uvs_view_f32 err_out, int32_t *status, int32_t *k_done, void *stream) {
    int L = 0;
    if (int rc = check_params(fp, T, &L)) return rc;
    if (!f.base || !dq.base || !x0.base) return fail(UVS_ERR_ARG, "%s", "f, dq and x0 views are required");
    if (fp->method == UVS_METHOD_MCKF) return fail(UVS_ERR_METHOD, "%s", "the single-precision replay runs KF, IMCCKF and GMCKF");
    auto v = [](const uvs_view_f32 &u) { return uvs::View32{u.base, u.trial_stride, u.step_stride, u.comp_stride}; };
    uvs::ReplayArgs32 A{*fp, T, v(f), v(dq), v(x0), v(x_out), v(err_out), status, k_done};
    if (!replay_f32(fp->m, fp->n, fp->method, T, (hipStream_t)stream, A))
        return fail(UVS_ERR_SHAPE, "%s", "the single-precision replay is instantiated for (m, n) = (8, 6) only");
    return check_launch("replay_f32_kernel");
}

int uvs_rmckf_step_f64(const uvs_filter_params *fp, int64_t T, double *X, double *P, const double *f, const double *f_old,
                       const double *dq_prev, int32_t first, int32_t k, double *dq_out, double *err_out, double *kappa_out,
                       int32_t *status, void *stream) {
    int L = 0;
    if (int rc = check_params(fp, T, &L)) return rc;
    if (!X || !P || !f || !f_old || !dq_prev || !dq_out || !err_out || !kappa_out || !status)
        return fail(UVS_ERR_ARG, "%s", "all step buffers are required");
    uvs::StepArgs A{*fp, T, X, P, f, f_old, dq_prev, first, k, dq_out, err_out, kappa_out, status};
    hipStream_t s = (hipStream_t)stream;
    // the single-step kernel selects the estimator at run time (MCKF included): at (8,6) its two-lane form spills to scratch, four lanes fit
    if (fp->lanes_per_filter == 0 && fp->m == 8 && fp->n == 6) L = 4;
    const bool launched = step_generic(fp->m, fp->n, L, T, s, A);
    if (!launched) return fail(UVS_ERR_SHAPE, "%s", "(m, n, lanes_per_filter) is not instantiated in libuvs_rmckf");
    return check_launch("step_kernel");
}

int uvs_stats_reduce_f64(int64_t T, int32_t K, int32_t m, uvs_view err, const double *t, const int32_t *k_done, double *stats,
                         void *stream) {
    if (T <= 0 || K < 0 || m <= 0 || !err.base || !t || !stats) return fail(UVS_ERR_ARG, "%s", "bad stats arguments");
    uvs_launch::stats((long long)T, K, m, uvs::to_view(err), t, k_done, stats, (hipStream_t)stream);
    return check_launch("stats_kernel");
}

int uvs_noise_generate_f64(const uvs_noise_params *np, int64_t T, const uint64_t *states, const double *zig, uvs_view out, void *stream) {
    if (!np || T <= 0 || !states || !zig || !out.base) return fail(UVS_ERR_ARG, "%s", "bad noise_generate arguments");
    if (np->m <= 0 || np->m % 2 || np->m > UVS_MAX_M || np->steps < 0) return fail(UVS_ERR_ARG, "%s", "noise: m must be even and <= UVS_MAX_M");
    if ((np->type & 0xff) < UVS_NOISE_WHITE || (np->type & 0xff) > UVS_NOISE_UNIFORM || (np->type & ~(0xff | UVS_NOISE_OPT_AS_WRITTEN)))
        return fail(UVS_ERR_ARG, "%s", "unknown noise type");
    noise(*np, (long long)T, (const unsigned long long *)states, zig, uvs::to_view(out), (hipStream_t)stream);
    return check_launch("noise_kernel");
}

int uvs_noise_generate_streams_f64(const uvs_noise_params *np, int64_t S, const uint64_t *states, const double *zig, double *out, int64_t stream_stride,
                                   int64_t step_stride, void *stream) {
    if (!np || S <= 0 || !states || !zig || !out || np->steps < 0) return fail(UVS_ERR_ARG, "%s", "bad noise_generate_streams arguments");
    const int base_type = np->type & 0xff;
    if ((base_type != UVS_NOISE_WHITE && base_type != UVS_NOISE_ALPHA_STABLE && base_type != UVS_NOISE_UNIFORM) || (np->type & ~(0xff | UVS_NOISE_OPT_AS_WRITTEN)))
        return fail(UVS_ERR_ARG, "%s", "noise streams: only the types with one generator per feature (WHITE_NOISE, ALPHA_STABLE, UNIFORM)");
    if (np->hold_cnt != 0) return fail(UVS_ERR_ARG, "%s", "noise streams: the outlier hold couples the two features of a pair; use uvs_noise_generate_f64");
    noise_streams(*np, (long long)S, (const unsigned long long *)states, zig, uvs::View{out, stream_stride, step_stride, 0}, (hipStream_t)stream);
    return check_launch("noise_streams_kernel");
}

int uvs_noise_kernel_variant(const uvs_noise_params *np) {
    if (!np || (np->type & 0xff) < UVS_NOISE_WHITE || (np->type & 0xff) > UVS_NOISE_UNIFORM || (np->type & ~(0xff | UVS_NOISE_OPT_AS_WRITTEN)))
        return fail(UVS_ERR_ARG, "%s", "bad noise parameters");
    return noise_variant(*np);
}

int uvs_pcg64_seed_u64(int64_t n, const uint64_t *seeds, uint64_t *states, void *stream) {
    if (n <= 0 || !seeds || !states) return fail(UVS_ERR_ARG, "%s", "bad pcg64_seed arguments");
    pcg64_seed((long long)n, (const unsigned long long *)seeds, (unsigned long long *)states, (hipStream_t)stream);
    return check_launch("pcg64_seed_kernel");
}

int uvs_debug_math_f64(int32_t which, int64_t n, const double *x, double *y, void *stream) {
    if (n <= 0 || !x || !y) return fail(UVS_ERR_ARG, "%s", "bad debug_math arguments");
    debug_math(which, (long long)n, x, y, (hipStream_t)stream);
    return check_launch("debug_math_kernel");
}

}  // extern "C"

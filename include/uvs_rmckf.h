/*
 * uvs_rmckf.h -- C ABI of libuvs_rmckf.so: batched RMCKF Jacobian estimator for
 * uncalibrated visual servoing on AMD MI355X (gfx950).
 *
 * The reference (AI-SPARC/uncalibrated-visual-servoing) has no native code and no
 * FFI: the estimator is ~35 inline numpy lines inside Experiment.run()
 * (experiment.py:164-300) and Monte-Carlo trials run sequentially from main.py:121-148.
 * This header is therefore the boundary a maintainer would bind with ctypes (see
 * INTEGRATION.md); every entry point cites the reference lines it replaces.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers owned by the caller; nothing is allocated,
 *     freed or synchronised inside the library (graph-capture safe).  Work is enqueued
 *     on the hipStream_t passed as `stream` (void* here; NULL = default stream).
 *   - Returns 0 on success, <0 on error (never throws); uvs_last_error() gives the text
 *     for the calling thread.  Thread-safe across distinct streams/devices; no global state.
 *   - fp64 throughout (numpy default).  Integers: int32 status/k_done.
 *   - Arrays indexed [trial][step][component] are passed as strided views so that the
 *     coalesced trial-fastest layout ([step][component][trial]) and the per-trial
 *     record layout ([trial][step][component]) are both expressible.  Strides are in
 *     elements (doubles).  A view with base == NULL disables that input/output.
 *   - m = 2*features (measurement rows), n = joints (regressor length), state X is
 *     m*n row-major (X.reshape(m, n), experiment.py:300), covariance is the m diagonal
 *     n x n blocks of the reference's mn x mn P (exactly block diagonal).
 */
#ifndef UVS_RMCKF_H
#define UVS_RMCKF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVS_MAX_M 32          /* measurement rows (16 features)            */
#define UVS_MAX_N 8           /* joints                                    */
#define UVS_MAX_POINTS 16

/* return codes */
#define UVS_OK 0
#define UVS_ERR_ARG (-1)      /* NULL/ill-formed argument                   */
#define UVS_ERR_SHAPE (-2)    /* (m, n, lanes_per_filter) not instantiated  */
#define UVS_ERR_HIP (-3)      /* HIP runtime error (see uvs_last_error)     */
#define UVS_ERR_METHOD (-4)   /* estimator not available on this path       */

/* experiment.py:6-11 (Method) and :13-15 (ExperimentStatus) */
enum { UVS_METHOD_ANALYTICAL = 1, UVS_METHOD_KF = 2, UVS_METHOD_MCKF = 3, UVS_METHOD_IMCCKF = 4, UVS_METHOD_GMCKF = 5 };
enum { UVS_STATUS_SUCCESS = 0, UVS_STATUS_FAIL = 1 };

/* Option bits of uvs_filter_params.reserved.
 * UVS_OPT_STRICT_PINV: numpy's pinv semantics (experiment.py:312: SVD, singular values <= 1e-15 sigma_max dropped) PROVEN on every control-law
 *   solve instead of watched for.  In the tuned QR kernels (RMCKF: lanes_per_filter 1 / 2 / 4 at (8,6) and (6,6); KF, IMCC-KF, MCKF: the default two lanes at (8,6) on the DH plant) every
 *   solve carries a certificate: cond_2(R) <= |R|_F |R^-1|_F, evaluated from the inverse of the triangular factor; below 2^42 nothing can be
 *   truncated, so the least-squares command IS pinv's; anything else marks the trial for the careful pass, which decides by a Jacobi SVD of the
 *   factor.  Costs 3-13 % of a launch (profiles/r06/strict_certificate_ab.txt; round 5: 16 x) -- cheap enough to audit the default mode's watches
 *   at full size (tests/test_gpu_rankdef.py::test_strict_pinv_audit_at_full_size: 65 536 trials, bit-identical).  Kernels without the certificate
 *   (the wide shape's normal equations, the generic templates, the replay) send every trial through the careful kernels instead, an order of
 *   magnitude slower.  Default off: the default mode WATCHES every solve -- QR kernels: a vanishing pivot, bad column scaling, a solution that
 *   grows by 2^34 against its right-hand side (catches the Kahan-like Jacobian of tests/golden/rankdef_gmckf_kahan_c1000, whose entries give
 *   nothing away); normal-equation kernels (the (32,7) closed loop, the replay's control wavefronts): the spread of the Cholesky pivots, a
 *   factorisation that breaks down, and (round 6) a refinement step that does not converge (correction >= 2^-20 of the solution), which is how a
 *   Kahan-like Jacobian shows there -- and re-runs only the trials it marks. */
#define UVS_OPT_STRICT_PINV 1
/* Small batches.  A closed-loop batch of the (8,6) shape that does not fill the chip (at most 16 384 trials; every estimator on the DH
 *   plant, lanes_per_filter == 0; MCKF too since round 5) runs with four lanes per filter instead of two -- half the trials per wavefront, twice the
 *   wavefronts, about a fifth less time per launch: what a rank of a strong-scaling split of main.py:121-148 wants.  These
 *   kernels form every sum in the two-lane kernel's order, so the RESULTS ARE BIT-IDENTICAL to the two-lane kernel's: the choice is invisible.
 * UVS_OPT_LATENCY: use the plain four-lane kernels there instead (their own summation order: a few % faster still, results that differ from the
 *   default mapping in the last bits; same oracle gates; not for MCKF). */
#define UVS_OPT_LATENCY 2
/* UVS_OPT_DIAG_DROP_SEG_FLAG (testing only): in a segmented launch the first segment of every trial chunk does not publish its hand-over
 *   counter, so the chunk's second segment runs out its spin budget (~65 ms) and takes the fallback -- recompute the trial from step 0 --
 *   that keeps segmented launches deadlock-free whatever the dispatcher does.  Results are bit-identical; tests/test_gpu_mckf_fpi.py forces it. */
#define UVS_OPT_DIAG_DROP_SEG_FLAG 4

/* Strided view of a [trial][step][component] array of doubles. */
typedef struct uvs_view {
    double *base;
    int64_t trial_stride, step_stride, comp_stride;
} uvs_view;

/* Estimator + control-law parameters: Experiment.__init__ (experiment.py:18-40) and the
 * constants of run() (:70-76, :119-122, :267-271, :280). */
typedef struct uvs_filter_params {
    int32_t m, n;               /* len(desired_f); joints (experiment.py:53-54)                    */
    int32_t method;             /* UVS_METHOD_*; GMCKF is the paper's RMCKF                         */
    int32_t annealing;          /* sigma_k = kernel_bw + anneal_span*(1 - k/k_max) (:267-271)       */
    int32_t k_max;              /* int(t_max/t_s) (:120)                                            */
    int32_t steps;              /* loop iterations K: number of t = t_s, 2 t_s, ... < t_max (:125)  */
    int32_t initial_guess;      /* 1: analytic interaction-matrix X0 (:86-114); 0: X0 from view     */
    int32_t lanes_per_filter;   /* 0 = library default (tuned kernels: 2 lanes at (8,6) and (6,6); 8 at (32,7) on */
                                /* the linear plant with x0 supplied, else the generic template with 16);        */
                                /* L > 0: L lanes cooperate on a filter (tuned kernel where one exists);         */
                                /* L < 0: generic template with |L| lanes (in-library cross-check of tuned code) */
    double kernel_bw;           /* sigma_0 (:37)                                                    */
    double anneal_span;         /* 100 (:271)                                                       */
    double gain;                /* ibvs_gain lambda (:26, :312)                                     */
    double dt;                  /* t_s (:24)                                                        */
    double reg;                 /* 0.001**2 added to Cy before inversion (:280)                     */
    double fpi_threshold;       /* MCKF fixed-point stop test (:38, :215)                           */
    int32_t fpi_epoch_max;      /* MCKF iteration cap; reaching it skips the correction (:39, :246) */
    int32_t reserved;           /* option bits, 0 = defaults.  bit 0: UVS_OPT_STRICT_PINV, bit 1: UVS_OPT_LATENCY, bit 2:  */
                                /* UVS_OPT_DIAG_DROP_SEG_FLAG (above).  bits 8-15: segments per MCKF trial for uvs_rmckf_closed_loop_ws_f64 (0 =       */
                                /* library's choice); all other bits must be 0                                             */
    double desired[UVS_MAX_M];  /* desired_f (:21)                                                  */
} uvs_filter_params;

/* Synthetic plant (the reference's real plant is an external CoppeliaSim process).
 *   UVS_PLANT_DH_PINHOLE: DH serial chain + pinhole camera on the last frame looking at fixed points.  Kinematics follow
 *     ur10_simulation.py:97-139,204-211; camera model and point placement are SURVEY.md Appendix A.
 *   UVS_PLANT_LINEAR: f = lin_f0 + lin_jacobian (q - lin_q0), a consistent linearised camera for shapes the DH model cannot
 *     provide (BASELINE config 5: m = 32, n = 7); the three arrays are DEVICE pointers (m*n row-major, m, n doubles). */
enum { UVS_PLANT_DH_PINHOLE = 0, UVS_PLANT_LINEAR = 1 };
typedef struct uvs_plant {
    int32_t n_joints, n_points;
    double theta_offset[UVS_MAX_N], d[UVS_MAX_N], a[UVS_MAX_N];
    double cos_alpha[UVS_MAX_N], sin_alpha[UVS_MAX_N];   /* filled by the host with its libm cos/sin(alpha) */
    double points[UVS_MAX_POINTS][3];                    /* world coordinates of the tracked points           */
    double focal, center;                                /* u = center + focal*x/z (experiment.py:97)         */
    int32_t kind, reserved;
    const double *lin_jacobian, *lin_f0, *lin_q0;
} uvs_plant;

/* Library identification. */
const char *uvs_version(void);
const char *uvs_last_error(void);
/* Lanes-per-filter variants compiled for (m, n): writes up to `cap` values, returns the count. */
int uvs_supported_lanes(int32_t m, int32_t n, int32_t *lanes, int32_t cap);

/*
 * Closed-loop Monte-Carlo batch: T independent servo trials, each the whole while-loop of
 * Experiment.run() (experiment.py:125-343) on the synthetic plant, one HIP grid.
 * Replaces main.py:121-148 (sequential trials) + experiment.py:48-359.
 *   q_start  [T][1][n]   in   start joints (main.py:129-134 jitter already applied)
 *   noise    [T][K][m]   in   per-step measurement noise (NoiseProfiler.getNoise(), noise.py:81-118); NULL = none
 *   x0       [T][1][m*n] in   initial state when !initial_guess
 *   x_out    [T][K][m*n] out  X after the update of step k                (NULL to skip)
 *   err_out  [T][K][m]   out  f - desired_f (error_log, experiment.py:327)
 *   q_out    [T][K][n]   out  joints at step k (q_log, :323)
 *   f_out    [T][K][m]   out  noisy features (f_log, :325)
 *   dq_out   [T][K][n]   out  commanded joint rate (:312)
 *   stats    [T][3]      out  ||ISE||_2, ||IAE||_2, ||ITAE||_2 over features (results/plot_errorbar.m:39-84)
 *   status   [T] int32   out  UVS_STATUS_* ; FAIL when X turns non-finite (pinv raises, :313-316).  REQUIRED: between the two
 *                             passes described below it also carries the marks of the trials to redo
 *   k_done   [T] int32   out  rows logged (k at exit, :345)
 *   x_final  [T][1][m*n], p_final [T][1][m*n*n] out  state after the last step (NULL to skip)
 *
 * numpy.linalg.pinv semantics (experiment.py:312: SVD, singular values <= 1e-15 * sigma_max dropped).  The kernels solve the control law
 * by Householder least squares (the (32,7) shape: normal equations with one refinement step), which equals pinv(J) y for full column rank.
 * Every solve is watched (see UVS_OPT_STRICT_PINV above for what each kind of kernel watches); a trial that trips a watch is marked and re-run from its first step by a second, careful kernel that the call enqueues
 * right behind the first: its control law finishes the QR with a Jacobi SVD of the n x n factor and applies numpy's cutoff, i.e. it
 * returns the truncated minimum-norm command the reference computes for a (numerically) rank-deficient Jacobian.  Healthy trials are not
 * touched by the second pass; the marks never leave the library.
 */
int uvs_rmckf_closed_loop_f64(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T,
                              uvs_view q_start, uvs_view noise, uvs_view x0,
                              uvs_view x_out, uvs_view err_out, uvs_view q_out, uvs_view f_out, uvs_view dq_out,
                              double *stats, int32_t *status, int32_t *k_done,
                              uvs_view x_final, uvs_view p_final, void *stream);

/*
 * The same call with a caller-owned scratch buffer (DEVICE memory, 8-byte aligned, contents irrelevant before and after the call; the
 * library still allocates nothing).  With it the MCKF closed loop (Method.MCKF, experiment.py:194-250) may cut every trial into segments
 * -- work items of a few dozen steps whose filter state crosses through the workspace -- so that a wavefront held up by trials whose
 * fixed-point iteration keeps iterating does not serialise with the next wavefront of its SIMD.  Results are
 * bit-identical to the call without a workspace.  uvs_rmckf_closed_loop_workspace_bytes() says how much this (fp, plant, T) wants: 0 when
 * segments would not help (other estimators, launches of one round or of many rounds of wavefronts); a NULL or too small workspace simply
 * runs unsegmented.  Bits 8-15 of fp->reserved, when non-zero, fix the number of segments (1 = never cut; testing / measurements).
 * One workspace must not be shared by calls that may run concurrently on different streams.
 */
size_t uvs_rmckf_closed_loop_workspace_bytes(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T);
/* Segments per trial the call above would use given a large enough workspace (1 = whole trials); host-side query for logs and benchmarks. */
int uvs_rmckf_closed_loop_segments(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T);
/* Lanes per filter the closed-loop call would use for this (fp, plant, T) -- lanes_per_filter, the shape's default, or 4 for a small batch
 * (see UVS_OPT_LATENCY above); 0 when the shape is not instantiated.  Host-side query. */
int uvs_rmckf_closed_loop_lanes(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T);
/* Health of a segmented launch.  A later segment that does not see its predecessor's hand-over within ~65 ms recomputes the trial from step 0
 * (results stay bit-identical; the path exists so that nothing ever hangs) -- silently, which on a GPU shared between processes or under a
 * debugger could turn a launch quadratic without anybody noticing.  The launch therefore counts such items: byte offset into the workspace
 * of an int32 that uvs_rmckf_closed_loop_ws_f64 zeroes and every fallen-back work item increments (0 = this launch is not segmented).  Read it
 * after the stream has finished; expected 0 (bench.py reports it, the tests assert it). */
size_t uvs_rmckf_closed_loop_fallback_offset(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T);
int uvs_rmckf_closed_loop_ws_f64(const uvs_filter_params *fp, const uvs_plant *plant, int64_t T,
                                 uvs_view q_start, uvs_view noise, uvs_view x0,
                                 uvs_view x_out, uvs_view err_out, uvs_view q_out, uvs_view f_out, uvs_view dq_out,
                                 double *stats, int32_t *status, int32_t *k_done,
                                 uvs_view x_final, uvs_view p_final, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Open-loop replay of recorded streams through the estimator + control law
 * (experiment.py:166-312 without the plant): the parity workhorse.
 *   f        [T][K+1][m]  in   observed features; row 0 is f_old of the first step (:128)
 *   dq       [T][K][n]    in   regressor of step k (previous command, :188); row 0 ignored (H = 0, :183)
 *   x0       [T][1][m*n]  in
 *   x_out    [T][K][m*n], err_out [T][K][m], kappa_out [T][K][m], dqcmd_out [T][K][n]  out (NULL to skip)
 *   status   [T] int32, k_done [T] int32  out   (status is REQUIRED when dqcmd_out is requested: same two-pass scheme as the closed loop)
 * Without dqcmd_out nothing couples a filter's rows and (8,6) runs the estimator-only kernel.  Fastest store paths, chosen from the views
 * (any other strides work, slower): x_out and err_out both as per-trial records (comp_stride 1, trial_stride m*n resp. m, i.e.
 * [step][trial][component]; T a multiple of 16, 16-byte aligned): whole 1 KB stores; trial-fastest ([step][component][trial]):
 * 512-byte stores for KF / RMCKF at lanes_per_filter = 0.
 */
int uvs_rmckf_replay_f64(const uvs_filter_params *fp, int64_t T, uvs_view f, uvs_view dq, uvs_view x0,
                         uvs_view x_out, uvs_view err_out, uvs_view kappa_out, uvs_view dqcmd_out,
                         int32_t *status, int32_t *k_done, uvs_view x_final, uvs_view p_final, void *stream);

/*
 * Single-precision estimator-only replay (SURVEY.md 8d "fp32 variant: report measured error"): the estimator of experiment.py:166-297 over
 * recorded streams with fp32 streams AND fp32 state, (8,6), KF / IMCC-KF / GMCKF.  A MEASURED LOWER-PRECISION VARIANT, not a drop-in: per-step
 * X deviates from the reference's fp64 runs by ~1e-6 relative (bounded at 1e-5 by tests/test_gpu_replay_f32.py), and no control law is
 * solved.  Views as in uvs_rmckf_replay_f64 with float elements (strides in floats); f has K + 1 rows, dq K rows, x0 one row.
 *   x_out [T][K][m*n], err_out [T][K][m] out (base NULL to skip); status / k_done [T] int32 out (NULL to skip; FAIL = non-finite X)
 */
typedef struct uvs_view_f32 {
    float *base;
    int64_t trial_stride, step_stride, comp_stride;
} uvs_view_f32;
int uvs_rmckf_replay_f32(const uvs_filter_params *fp, int64_t T, uvs_view_f32 f, uvs_view_f32 dq, uvs_view_f32 x0,
                         uvs_view_f32 x_out, uvs_view_f32 err_out, int32_t *status, int32_t *k_done, void *stream);

/*
 * One estimator + control step for T filters whose state lives in HBM between calls: what
 * Experiment.run() does between getCameraImage() and setJointsPos() (experiment.py:166-312) when
 * the robot is external (live simulator).  All arrays contiguous, trial-major.
 *   X [T][m*n] inout, P [T][m][n][n] inout, f [T][m], f_old [T][m], dq_prev [T][n] in
 *   first: 1 on the first iteration (H = 0, :183-184); k: loop index for annealing (:270)
 *   dq_out [T][n], err_out [T][m], kappa_out [T][m] out; status [T] int32 out (FAIL = non-finite X)
 * numpy's pinv semantics inline (no second pass follows a single step): the plain least-squares solve with the default mode's watches, and the
 * careful solve (QR finished by an SVD of the factor, numpy's cutoff) for a filter whose watch fires -- for every filter under UVS_OPT_STRICT_PINV.
 * The per-call operands (f, f_old, dq_prev, dq_out, err_out, kappa_out, status) may live in pinned host memory mapped to the device: the kernel reads
 * and writes them in place, which is how the package's drop-in route runs without a copy in either direction (engine.FilterBank.step_host).
 */
int uvs_rmckf_step_f64(const uvs_filter_params *fp, int64_t T, double *X, double *P, const double *f,
                       const double *f_old, const double *dq_prev, int32_t first, int32_t k,
                       double *dq_out, double *err_out, double *kappa_out, int32_t *status, void *stream);

/*
 * Per-trial ISE/IAE/ITAE norms from an error trajectory (results/plot_errorbar.m:39-84).
 *   err [T][K][m] in, t [K] in (device), k_done [T] in (NULL = K rows), stats [T][3] out
 */
int uvs_stats_reduce_f64(int64_t T, int32_t K, int32_t m, uvs_view err, const double *t, const int32_t *k_done,
                         double *stats, void *stream);

/* noise.py:7-12 (NoiseType) */
enum { UVS_NOISE_WHITE = 1, UVS_NOISE_GAUSSIAN_MIXTURE = 2, UVS_NOISE_GAUSSIAN_BIMODAL = 3, UVS_NOISE_ALPHA_STABLE = 4, UVS_NOISE_UNIFORM = 5 };
/* Option bit of uvs_noise_params.type (the low byte is the noise type).  UVS_NOISE_OPT_AS_WRITTEN: evaluate the general Chambers-Mallows-Stuck
 * branches of ALPHA_STABLE the way noise.py:188-199 writes them -- library sin / cos / log, two library pow(), the reference's order of
 * operations -- instead of folding the two powers into one exponential.  The default kernels agree with numpy to <= 5e-13 relative (up to a
 * few dozen ulp in the far tails, where the folded exponent is large); this variant stays within a few ulp everywhere (numpy's own scalar /
 * SIMD spread is <= 2 ulp), at the generator cost recorded in profiles/r06/noise_as_written.txt.  Other types and the special cases of
 * ALPHA_STABLE (alpha = 2, Cauchy, Levy, alpha = 1 with skew) already call the library functions and ignore the bit. */
#define UVS_NOISE_OPT_AS_WRITTEN 0x100

/* Parameters of NoiseProfiler (noise.py:31-79).  The derived fields are filled by the host with the same Python float
 * arithmetic the reference uses, so that no rounding differs: inv_alpha = 1/alpha, expo = (1-alpha)/alpha,
 * cms_const = beta*tan(pi*alpha/2), cms_B = arctan(cms_const), cms_S = (1+cms_const^2)^(1/(2 alpha)) (noise.py:191-196),
 * shift = (2/pi)*beta*gamma*log(gamma) (noise.py:203), sqrt2 = sqrt(2) (noise.py:181), two_over_pi = 2/pi (noise.py:199). */
typedef struct uvs_noise_params {
    int32_t type, m, steps, hold_cnt;       /* hold_cnt = noise_hold ? noise_hold_cnt : 0 */
    double std, mean, rho;
    double alpha, beta, gamma, delta;
    double inv_alpha, expo, one_minus_alpha, cms_const, cms_B, cms_S, shift, sqrt2, two_over_pi;
} uvs_noise_params;

/*
 * numpy.random.PCG64(seed) for n integer seeds (< 2^64) at once, on the device: states[i] = (state_hi, state_lo, inc_hi, inc_lo)
 * as SeedSequence + pcg64_set_seed produce them (numpy/random/bit_generator.pyx, src/pcg64); replaces constructing
 * Generator(PCG64(seed + 10*i)) / PCG64(2*seed + i) one by one (noise.py:55-70).  seeds, states: device pointers.
 */
int uvs_pcg64_seed_u64(int64_t n, const uint64_t *seeds, uint64_t *states, void *stream);

/*
 * Noise streams of T trials generated on the device: out[t][k][i] is the k-th getNoise() value of feature i of
 * NoiseProfiler(m, type, seed_t, ...) (noise.py:81-118), from the states of the trial's generators (uvs_pcg64_seed_u64 above):
 *   states [T][n_gen][4] uint64 (device) = (state_hi, state_lo, inc_hi, inc_lo) of PCG64(seed_t + 10*j) for j < gens*m
 *          (noise.py:66-70) followed, for the mixtures, by PCG64(2*seed_t + i) for i < m (noise.py:55-59);
 *   zig    768 doubles (device): numpy's ziggurat tables fi[256], wi[256], ki[256] (ki as raw uint64 bits).
 * Uniform, normal and mixture streams reproduce numpy bit for bit (tail samples of the normal to 1-2 ulp); Cauchy and the
 * Chambers-Mallows-Stuck transforms agree to <= 5e-13 relative (device libm vs host libm; by default the CMS powers are folded into one
 * exponential, which costs up to a few dozen ulp in the far tails -- UVS_NOISE_OPT_AS_WRITTEN above evaluates them as the reference writes them).
 */
int uvs_noise_generate_f64(const uvs_noise_params *np, int64_t T, const uint64_t *states, const double *zig, uvs_view out, void *stream);

/*
 * The same streams without their redundancy.  NoiseProfiler seeds generator i of a trial with seed + 10 i (noise.py:70) and the Monte-Carlo
 * driver gives trial t the seed seed0 + t (main.py:137-139): for the types with one generator per feature (WHITE_NOISE, ALPHA_STABLE, UNIFORM)
 * and no outlier hold, feature i + 1 of trial t is the very stream of feature i of trial t + 10, so T consecutive trials hold only
 * S = T + 10 (m - 1) distinct streams.  This call generates stream s = the getNoise() sequence of ONE generator in state states[s]
 * (PCG64(seed0 + s), uvs_pcg64_seed_u64) into out[s * stream_stride + k * step_stride], k < np->steps (np->m is ignored, np->hold_cnt must
 * be 0); the closed loop then reads the noise of trial t, feature i through the view {out, trial_stride = stream_stride,
 * step_stride, comp_stride = 10 * stream_stride}.  Values are bit-identical to uvs_noise_generate_f64's: 1/m of the generator work and of
 * the buffer (BASELINE config 2: 0.16 GB instead of 1.25 GB per cell).
 */
int uvs_noise_generate_streams_f64(const uvs_noise_params *np, int64_t S, const uint64_t *states, const double *zig,
                                   double *out, int64_t stream_stride, int64_t step_stride, void *stream);

/*
 * Which instantiation uvs_noise_generate_f64 launches for these parameters (host-side query, no GPU work): 0 = the kernel of np->type,
 * 1 = the alpha-stable kernel specialised for beta = 0 (noise.py:188-191), which evaluates cos((1 - alpha) V) by the addition theorem;
 * it is selected only where that costs < 2e-14 relative (alpha from about 0.1 to 1.999, not 1 or 2); 2 = the as-written kernel
 * (UVS_NOISE_OPT_AS_WRITTEN on a general Chambers-Mallows-Stuck parameter set).  <0 on bad arguments.
 */
int uvs_noise_kernel_variant(const uvs_noise_params *np);

/*
 * Test hook: evaluates the library's fp64 helper functions on the device so that tests can bound their error against
 * numpy.  which: 0 fast reciprocal, 1 sqrt, 2 rsqrt, 3 sin, 4 cos (bounded-argument sincos with library fallback), 5 exp,
 * 6 exp for non-positive arguments, 7 log (own routine on normal positive arguments, library elsewhere), 8 exp with clamped argument,
 * 9 / 10 sin / cos of 0.7 + x by the short angle-addition polynomials (|x| <= 0.1), 11 / 12 by the long ones (|x| <= 1).
 */
int uvs_debug_math_f64(int32_t which, int64_t n, const double *x, double *y, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UVS_RMCKF_H */

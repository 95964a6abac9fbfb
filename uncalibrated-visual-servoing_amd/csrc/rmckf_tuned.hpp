// Tuned closed-loop kernel: 1, 2 or 4 lanes per filter -- the headline path (BASELINE config 2 runs it with L = 2).
//
// Sizing.  65 536 trials x (8 blocks x 21 doubles of P) is 88 MB -- 69 % of the chip's whole VGPR+AGPR file
// (1024 SIMDs x 512 regs x 64 lanes x 4 B) and twice its LDS -- so P lives in registers, one (L <= 2) or two (L = 4)
// wavefronts fit per SIMD and nothing hides latency: a scratch spill is a full memory round trip on the critical path
// (measured: 250 scratch loads per step cost 7x the arithmetic).  Only 256 of the 512 registers are VALU-addressable; the
// AGPR half and LDS are parking space at ~20 cycles per double and round trip.  With one wavefront per SIMD every
// instruction of any kind costs a ~4.5-cycle issue slot, so the design minimises instruction count and branches:
//   * L lanes share a filter, lane s owning rows s, s+L, s+2L, ... (interleaved, so all lanes run the same static
//     Householder code and the (u, v) rows of a point fall on even / odd lanes);
//   * L = 2: half of the lane's covariance blocks stay in VGPRs, the other half, X and the ISE/IAE/ITAE accumulators live in
//     LDS as [component][lane] (conflict free); L = 4: everything but the accumulators is in VGPRs;
//   * group exchanges (partial sums, pivots, kinematic partial products) are DPP quad_perm moves -- no LDS, no ds_bpermute;
//   * the kinematic chain is split over the lanes (3 + 3 links for L = 2, 2 + 2 + 2 for L = 4), each lane tracks only its joints;
//   * plant constants are broadcast from LDS instead of occupying ~90 SGPRs that would spill to VGPR lanes;
//   * per-lane stream cursors advance by uniform strides; the step body is straight-line code with unconditional stores;
//   * divisions / roots / sincos come from rmckf_math.hpp (v_rcp/v_rsq + Newton steps, bounded-argument sincos).
// With the trial-fastest layout ([step][component][trial]) each wavefront store covers 512 / L contiguous bytes per owned row.
#pragma once
#include <type_traits>
#include "rmckf_device.hpp"
#include "rmckf_math.hpp"

namespace uvs {

// ---- Diagnostic builds (`make stamps`, `make quick QDEF=-DUVS_...`; into tools/diag/, never the shipped library).  One switch each; the
// kernel below reads them with `if constexpr`, so a build without them contains none of this code.  Each writes its clock sums over the first
// words of a wavefront's / segment's slice of `stats` (garbage in that build) and is read with tools/read_stamps.py / tools/wave_times.py:
//   UVS_STAMPS       per-phase cycle sums (s_memtime) of the step loop            UVS_FPI_STAMPS   cycles of the MCKF fixed-point branch by phase
//   UVS_ITEM_STAMPS  where a work item's time goes (entry / state / steps / hand-over, 100 MHz clock)
//   UVS_WAVE_TIMES   when and where every wavefront ran (100 MHz clock, HW_ID, XCC_ID)
#ifdef UVS_STAMPS
constexpr bool kDiagStamps = true;
#else
constexpr bool kDiagStamps = false;
#endif
#ifdef UVS_FPI_STAMPS
constexpr bool kDiagFpi = true;
#else
constexpr bool kDiagFpi = false;
#endif
#ifdef UVS_ITEM_STAMPS
constexpr bool kDiagItems = true;
#else
constexpr bool kDiagItems = false;
#endif
#ifdef UVS_WAVE_TIMES
constexpr bool kDiagWaves = true;
#else
constexpr bool kDiagWaves = false;
#endif

UVS_DEV unsigned long long diag_cycles() {                       // shader clock
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
UVS_DEV unsigned long long diag_ticks() {                        // constant 100 MHz clock, one for the whole device
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
struct DiagPhases {                                              // sums of cycles between consecutive stamps, by slot
    unsigned long long sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
    UVS_DEV void stamp(int slot) {                               // slot < 0: restart the interval without booking it
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = diag_cycles();
        __builtin_amdgcn_sched_barrier(0);
        if (slot >= 0) sum[slot] += now - last;
        last = now;
    }
};
#define UVS_STAMP(slot) do { if constexpr (kDiagStamps) diag_steps.stamp(slot); } while (0)
#define UVS_FPI_STAMP(slot) do { if constexpr (kDiagFpi) diag_fpi.stamp(slot); } while (0)

// DPP quad_perm move of a double (two 32-bit moves).  CTRL = a | b<<2 | c<<4 | d<<6 selects the source lane of each lane of a quad.
template <int CTRL>
UVS_DEV double dpp_quad(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int kSwapPair = 0xB1;      // quad_perm [1,0,3,2]: partner lane
constexpr int kFromEven = 0xA0;      // quad_perm [0,0,2,2]: value of the pair's even lane
constexpr int kFromOdd = 0xF5;       // quad_perm [1,1,3,3]: value of the pair's odd lane

constexpr int kSwapHalf = 0x4E;      // quad_perm [2,3,0,1]: the other pair of the quad
// Explicit parking of a double in two AGPRs (the "a" constraint keeps the halves in accumulator registers between put and get).  For state that
// is live THROUGH a register-hungry rare branch: left to itself the allocator keeps such state in VGPRs and runs the branch's own arrays out of
// AGPRs, one v_accvgpr move per use (measured on the MCKF fixed-point branch: 700 moves per firing against 168 with the covariance blocks parked).
struct ParkedDouble { int lo, hi; };
UVS_DEV ParkedDouble agpr_park(double v) {
    ParkedDouble a;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a.lo) : "v"(__double2loint(v)));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a.hi) : "v"(__double2hiint(v)));
    return a;
}
UVS_DEV double agpr_unpark(const ParkedDouble &a) {
    int lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(a.lo));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(a.hi));
    return __hiloint2double(hi, lo);
}
// DPP row_shr:SH of a double: lane i of a 16-lane row receives lane i - SH (0.0 where that falls out of the row)
template <int SH>
UVS_DEV double dpp_row_shr(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + SH, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + SH, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Sum over the L (1, 2 or 4) lanes of a filter; every lane gets the bit-identical total.
template <int L>
UVS_DEV double pair_sum(double v) {
    if constexpr (L == 1) return v;
    v += dpp_quad<kSwapPair>(v);
    if constexpr (L == 4) v += dpp_quad<kSwapHalf>(v);
    return v;
}
// Value held by lane OWNER of the group, delivered to all its lanes.
template <int L, int OWNER>
UVS_DEV double pair_from(double v) {
    if constexpr (L == 1) return v;
    if constexpr (L == 2) return OWNER ? dpp_quad<kFromOdd>(v) : dpp_quad<kFromEven>(v);
    return dpp_quad<OWNER * 0x55>(v);                   // quad_perm [o,o,o,o]
}
template <int L>
UVS_DEV double pair_from_dyn(double v, int owner) {      // owner is a compile-time constant after unrolling
    if constexpr (L == 1) return v;
    switch (owner) {
        case 0: return pair_from<L, 0>(v);
        case 1: return pair_from<L, 1>(v);
        case 2: return pair_from<L, (L > 2 ? 2 : 0)>(v);
        default: return pair_from<L, (L > 2 ? 3 : 0)>(v);
    }
}
// Pin a value in a VGPR.  Without it LLVM folds "cond ? a[1] : a[0]" on a register-resident array into a variably indexed
// access, which on AMDGPU means: spill the array to scratch and load it back through memory -- per lane, per step.
UVS_DEV double in_reg(double x) {
    asm volatile("" : "+v"(x));
    return x;
}
// Per-lane choice among the L values v[0..L) by the lane's position in its group.
template <int L>
UVS_DEV double pick_sub(const double *v, int sub) {
    if constexpr (L == 1) return v[0];
    if constexpr (L == 2) return sub ? in_reg(v[1]) : in_reg(v[0]);
    const double lo = (sub & 1) ? in_reg(v[1]) : in_reg(v[0]), hi = (sub & 1) ? in_reg(v[3]) : in_reg(v[2]);
    return (sub & 2) ? hi : lo;
}

// One row of the block-form estimator on a lane's registers: predict P_i + Q, innovation, correntropy weight, gain,
// state update, rank-1 Joseph downdate.  x: row i of X; pb: its packed covariance block; dq: the regressor (the previous command);
// zi: the row's measurement f_i - f_old_i.  chk accumulates 0 * x so that it turns NaN as soon as an entry of X is non-finite.
// Shared by the tuned closed-loop kernel and both tuned replay kernels.
// What the first pass of the fixed-point MCKF (experiment.py:194-250) leaves for the convergence test ||Xc - X|| / ||X|| <= fpi_threshold
// (:244, norms over ALL rows of the filter): the lane's share of both squared norms, and whether a correntropy weight underflowed to 0
// (inv(Cy) raises in the reference and the correction is skipped, :231-236).
struct FpiProbe {
    double num = 0.0, den = 0.0;                                 // num = +inf: not decidable here, leave the trial to the careful pass
    bool skip = false;                                           // a weight Cy underflowed to 0: the whole correction of this step is skipped
    bool poison = false;                                         // a weight of one of the lane's rows is subnormal: NaN gain, the trial FAILs
    bool unsure = false;                                         // ... or sits so close to the underflow that only the careful pass may decide
    double row_gamma = 0.0, row_a = 0.0, row_nu = 0.0;           // first-pass gain, h.P h and innovation of the row just processed
    double row_s2 = 0.0, row_gg = 0.0;                           // (gamma nu)^2 and |g|^2 of that row: the terms of num, for kernels that sum them in another lane order (EMU2)
    bool known = false;                                          // the caller already holds this row's innovation and weight argument (its pre-pass formed them:
    double known_nu = 0.0, known_arg = 0.0;                      // the same operations on the same values) -- the row does not form them again
};
// exp(x) == 0.0 in fp64 exactly when x < ln(2^-1075) = -745.1332191019412076...  The argument itself carries a few ulp of rounding
// (1.6e-13 absolute) that differ between this arithmetic and numpy's, so within 1e-11 of the boundary the tuned kernels do not decide
// themselves but mark the trial for the second pass.  (Round 2 used a band of +-0.005: at alpha = 1 that marked ~8 of 65 536 trials per
// sweep, and the eight lone wavefronts of the second pass took three times as long as the whole first pass.)
constexpr double kExpZeroBelow = -745.1332191019512, kExpNonzeroAbove = -745.1332191019312;
// A weight that is subnormal but not 0 -- at most 2^-1024, so that its reciprocal overflows -- does not make inv(Cy) raise: it returns inf,
// the reference's dense product Br @ inv(Cy) @ Br.T (experiment.py:232) turns 0 * inf into NaN, and the gain, the state and the trial are
// lost (pinv raises in the control law, :312-316: ExperimentStatus.FAIL at this step).  With Cauchy-like noise and the reference's shipped
// parameters that is how ~7 % of its alpha = 1 trials end (innovations between 37.7 and 38.6 sigma; fixtures tests/golden/fpi_*_fail,
// fpi_default_*).  The kernels reproduce it by poisoning the gain of the row.
constexpr double kRcpOverflowsAtOrBelow = 0x1p-1024;
constexpr double kRcpOverflowArg = -709.78271289338397;          // ln(2^-1024): the same boundary on the argument of the exponential
constexpr double kExpArgBand = 1e-11;                            // |argument - boundary| within which the tuned kernels hand the verdict to the careful pass
UVS_DEV double mckf_poison(double gain, double cy) { return (cy <= kRcpOverflowsAtOrBelow) ? __builtin_nan("") : gain; }

// Pre-pass of an MCKF step over the lane's rows: innovation of every row against the prior state, to find out whether some Cy is
// exactly 0 -- inv(Cy) then raises in the reference and the step keeps only the prediction (experiment.py:225-236; with Cauchy-like noise
// that is 1-2 % of the steps, it is not an exotic path).  `probe(r)` returns nu_r^2 * (-1 / (2 sigma^2)), the argument of the weight.
template <int R, typename ArgOfRow>
UVS_DEV void mckf_underflow_prepass(FpiProbe &fpi, ArgOfRow arg_of_row) {
    bool zero = false, unsure = false, poison = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double a = arg_of_row(r);
        zero |= a < kExpZeroBelow;
        unsure |= (a >= kExpZeroBelow) && (a <= kExpNonzeroAbove);    // (a NaN argument is not "unsure": that trial FAILs by itself, no second pass)
        poison |= a < kRcpOverflowArg;                                // decided on the argument: no weight has to stay live for it
        unsure |= fabs(a - kRcpOverflowArg) <= kExpArgBand;           // ... and within rounding of that boundary only the careful pass (which decides on the weight) may say
    }
    fpi.skip = zero;
    fpi.unsure = unsure;
    fpi.poison = poison;                                              // (a zero weight anywhere in the filter wins: inv(Cy) raises first)
}
// The reference's NaN state after a subnormal weight (see kRcpOverflowsAtOrBelow), applied to the finiteness probe of the step: the trial
// FAILs at this step exactly as if its X had turned NaN; what the rows computed instead (a gain of ~0) lands in rows at and after k_done,
// which are unspecified.  skip and poison must already be filter-wide (summed over the lanes of the filter).
UVS_DEV double mckf_poisoned(const FpiProbe &f, double chk) { return (f.poison && !f.skip) ? __builtin_nan("") : chk; }

// Verdict after the rows of a step (num / den summed over the lanes of the filter): true when the first pass is not the whole story --
// a second fixed-point pass would run, the correction would be skipped, or the test is too close to call in different rounding.  The
// tuned kernels then mark the trial and the careful second pass (generic template, full fixed-point iteration) redoes it.  On the
// reference's own configuration (threshold 0.1) every step of every fixture converges in the first pass.
UVS_DEV bool fpi_needs_more(const FpiProbe &f, const uvs_filter_params &fp) {
    const double thr2 = fp.fpi_threshold * fp.fpi_threshold;
    return f.unsure || fp.fpi_epoch_max <= 1 || !(f.num <= thr2 * f.den * (1.0 - 1e-9));
}

// ---- second and later fixed-point passes of the MCKF (experiment.py:215-245), one row (block) at a time on a lane's registers.
// Undo of the optimistic first-pass commit of rmckf_row: from (x_new, P_new) and the row's gamma, a, nu back to the prior x and the
// predicted block P + Q, and the first-pass gain row k1 = gamma (P + Q) h.  P_new h = g (1 - beta a) gives g without a second copy of P.
template <int N>
UVS_DEV void mckf_undo_row(double (&x)[N], double (&pb)[Sym<N>::NP], const double (&h)[N], double gamma, double a, double nu, double (&k1)[N]) {
    const double beta = gamma * (2.0 - gamma * (a + 1.0));
    const double inv = fast_rcp(1.0 - beta * a);                // (1 - gamma a)^2 + gamma^2 a > 0
    double g[N];
#pragma unroll
    for (int l = 0; l < N; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) acc = fma(pb[Sym<N>::at(l, j)], h[j], acc);
        g[l] = acc * inv;
    }
#pragma unroll
    for (int l = 0; l < N; ++l) {
        const double w = beta * g[l];
#pragma unroll
        for (int j = l; j < N; ++j) pb[Sym<N>::at(l, j)] = fma(w, g[j], pb[Sym<N>::at(l, j)]);
        k1[l] = g[l] * gamma;
        x[l] = fma(-k1[l], nu, x[l]);
    }
}
// Lower Cholesky factor of a row's predicted block, packed; the diagonal keeps 1 / L_jj and ljj[] the L_jj themselves.  The predicted block
// does not change between the passes of a step, so a kernel that keeps a row on one lane factors it once per step (round 4 refactored it every pass).
template <int N>
UVS_DEV void mckf_factor_row(const double (&pp)[Sym<N>::NP], double (&Lc)[Sym<N>::NP], double (&ljj)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double dsum = pp[Sym<N>::at(j, j)];
#pragma unroll
        for (int k2 = 0; k2 < j; ++k2) dsum = fma(-Lc[Sym<N>::at(k2, j)], Lc[Sym<N>::at(k2, j)], dsum);
        double lj, rl;
        fast_sqrt_rsqrt(dsum, lj, rl);
        Lc[Sym<N>::at(j, j)] = rl;                               // the diagonal keeps 1 / L_jj: only reciprocals of it are ever needed
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = pp[Sym<N>::at(j, i)];
#pragma unroll
            for (int k2 = 0; k2 < j; ++k2) v = fma(-Lc[Sym<N>::at(k2, i)], Lc[Sym<N>::at(k2, j)], v);
            Lc[Sym<N>::at(j, i)] = v * rl;
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) ljj[j] = fast_rcp(Lc[Sym<N>::at(j, j)]);       // L_jj back from its reciprocal
}
// One further pass for one row: current iterate xc = x + k nu0 -> new gain row kn (same arithmetic as Rows::update_mckf in
// rmckf_device.hpp, which the careful / generic kernels run).  dd[l] = xn[l] - xc[l] and xcv[l] = xc[l] are this row's terms of
// ||xn - xc||^2 and ||xc||^2 (the caller sums them in the filter's row order); bad: a weight Cy is 0.
template <int N>
UVS_DEV void mckf_iterate_row(const double (&x)[N], const double (&Lc)[Sym<N>::NP], const double (&ljj)[N], const double (&h)[N], double zi,
                              double neg_half_inv_s2, const double (&k)[N], double (&kn)[N], double (&dd)[N], double (&xcv)[N], bool &bad) {
    double nu0 = zi, xc[N], ex[N], t[N], g[N];
#pragma unroll
    for (int j = 0; j < N; ++j) nu0 = fma(-x[j], h[j], nu0);    // prior innovation (the gain is applied to it, experiment.py:242)
#pragma unroll
    for (int j = 0; j < N; ++j) xc[j] = fma(k[j], nu0, x[j]);
#pragma unroll
    for (int i = 0; i < N; ++i) {                                // L ex = x - xc
        double v = x[i] - xc[i];
#pragma unroll
        for (int k2 = 0; k2 < i; ++k2) v = fma(-Lc[Sym<N>::at(k2, i)], ex[k2], v);
        ex[i] = v * Lc[Sym<N>::at(i, i)];
    }
    double ez = zi;
#pragma unroll
    for (int j = 0; j < N; ++j) ez = fma(-xc[j], h[j], ez);
    const double cy = exp_nonpos((ez * ez) * neg_half_inv_s2);
    bad |= (cy == 0.0);
#pragma unroll
    for (int j = 0; j < N; ++j) {                                // t = Cx^-1 L^T h
        double v = ljj[j] * h[j];
#pragma unroll
        for (int i = j + 1; i < N; ++i) v = fma(Lc[Sym<N>::at(j, i)], h[i], v);
        t[j] = v * fast_rcp(exp_nonpos((ex[j] * ex[j]) * neg_half_inv_s2));    // a Cx of 0 gives inf -> NaN state -> the trial FAILs
    }
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {                                // g = L t = P_hat h
        double v = ljj[i] * t[i];
#pragma unroll
        for (int j = 0; j < i; ++j) v = fma(Lc[Sym<N>::at(j, i)], t[j], v);
        g[i] = v;
        a = fma(h[i], v, a);
    }
    const double gain = mckf_poison(cy * fast_rcp(fma(a, cy, 1.0)), cy);     // cy == 0 is `bad` (the caller skips the correction), not poison
#pragma unroll
    for (int l = 0; l < N; ++l) {
        kn[l] = g[l] * gain;
        dd[l] = fma(kn[l], nu0, x[l]) - xc[l];
        xcv[l] = xc[l];
    }
}
// Final state of a row after the iteration: x + k nu0 and the Joseph form with a gain row that is no longer gamma (P + Q) h
// (experiment.py:297): P - k g^T - g k^T + (h.g + 1) k k^T, g = (P + Q) h.
template <int N>
UVS_DEV void mckf_commit_row(double (&x)[N], double (&pp)[Sym<N>::NP], const double (&h)[N], double zi, const double (&k)[N], double &chk) {
    double nu0 = zi, g[N], a = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) nu0 = fma(-x[j], h[j], nu0);
#pragma unroll
    for (int l = 0; l < N; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) acc = fma(pp[Sym<N>::at(l, j)], h[j], acc);
        g[l] = acc;
        a = fma(h[l], acc, a);
    }
    const double c2 = a + 1.0;
#pragma unroll
    for (int l = 0; l < N; ++l) {
#pragma unroll
        for (int j = l; j < N; ++j) {
            double v = pp[Sym<N>::at(l, j)];
            v = fma(-k[l], g[j], v);
            v = fma(-g[l], k[j], v);
            v = fma(c2 * k[l], k[j], v);
            pp[Sym<N>::at(l, j)] = v;
        }
        x[l] = fma(k[l], nu0, x[l]);
        chk = fma(x[l], 0.0, chk);
    }
}

// Hook: a kernel may hand the row update work that is independent of it -- the streaming stores of values finished earlier -- to be
// issued at N + 1 fixed points spread over the row's arithmetic (hook(integral_constant<int, i>), i = 0..N), each pinned between
// scheduling fences.  A wavefront that issues its stores in one burst stalls at the full store queue while the SIMD has nothing else to run;
// one store every ~20 arithmetic instructions keeps both busy.  NoHook (every other kernel): nothing is emitted, the schedule is hipcc's.
struct NoHook {};
template <int I, typename Hook>
UVS_DEV void row_hook(Hook &hook) {
    if constexpr (!std::is_same<Hook, NoHook>::value) {
        __builtin_amdgcn_sched_barrier(0);
        hook(std::integral_constant<int, I>{});
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Share: KF and IMCC-KF weigh every row of a filter alike (gain factor 1 / (a + 1) resp. c / (c a + 1) with ONE c), so their covariance
// blocks stay identical for all rows: P_i <- P_i + I - beta g g^T with g = (P_i + I) h, and beta a function of h^T g alone.  A kernel may
// keep one block per lane: the leading row runs the full update and leaves g and the gain factor in a RowShare, the other rows of the lane
// only move their x (rmckf_row_follow: 20 instructions instead of 127).  Same operations on the same values: bit-identical results.
struct NoShare {};
template <int N>
struct RowShare { double g[N]; double gamma; };

template <int N, int METHOD, typename Hook, typename Share>
UVS_DEV void rmckf_row(double (&x)[N], double (&pb)[Sym<N>::NP], const double (&dq)[N], double zi, double neg_half_inv_s2, double c_shared,
                       double reg, double &kap, double &chk, FpiProbe &fpi, Hook &hook, Share &share) {
    static_assert(std::is_same<Hook, NoHook>::value || N == 6, "hook points are placed for n = 6");
    double g[N];
    double pred = 0.0;
    row_hook<0>(hook);
    double nu;
    if (METHOD == UVS_METHOD_MCKF && fpi.known) {
        nu = fpi.known_nu;
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) pred = fma(x[j], dq[j], pred);
        nu = zi - pred;                                          // innovation (experiment.py:274)
    }
#pragma unroll
    for (int l = 0; l < N; ++l) pb[Sym<N>::at(l, l)] += 1.0;     // P + Q (experiment.py:167)
#pragma unroll
    for (int l = 0; l < N; ++l) {
        double acc = pb[Sym<N>::at(l, 0)] * dq[0];
#pragma unroll
        for (int j = 1; j < N; ++j) acc = fma(pb[Sym<N>::at(l, j)], dq[j], acc);
        g[l] = acc;
        if (l == 1) row_hook<1>(hook);
        if (l == 3) row_hook<2>(hook);
    }
    row_hook<3>(hook);
    double a = 0.0;
#pragma unroll
    for (int l = 0; l < N; ++l) a = fma(dq[l], g[l], a);
    double gamma;
    if constexpr (METHOD == UVS_METHOD_GMCKF) {
        kap = exp_nonpos((nu * nu) * neg_half_inv_s2);           // utils.py:171-172
        const double d = kap + reg;                              // gamma = 1 / (a + 1/d) = d / (a d + 1) (experiment.py:280-286)
        gamma = d * fast_rcp(fma(a, d, 1.0));
    } else if constexpr (METHOD == UVS_METHOD_MCKF) {
        // first fixed-point pass: Xc = X, so Cx = I and P_hat = P; gain = 1 / (a + 1 / Cy) (experiment.py:225-242).  The state update
        // and the Joseph form below are then exactly those of the other estimators; kappa of the control law is 1 (:303-308)
        const double cy = exp_nonpos(fpi.known ? fpi.known_arg : (nu * nu) * neg_half_inv_s2);
        // skipped correction: X stays, P keeps the prediction (gamma = 0 below).  (A subnormal weight -- fpi.poison -- is not injected here:
        // a select on the gain costs this register-bound kernel 108 B of scratch; the caller FAILs the trial through mckf_poisoned.)
        gamma = fpi.skip ? 0.0 : cy * fast_rcp(fma(a, cy, 1.0));
        kap = 1.0;
        double gg = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) { gg = fma(g[j], g[j], gg); fpi.den = fma(x[j], x[j], fpi.den); }
        const double s = gamma * nu;
        fpi.num = fma(s * s, gg, fpi.num);                       // ||K (Z - H X)||^2 of this row (0 when skipped: no second pass then)
        fpi.row_s2 = s * s;
        fpi.row_gg = gg;
        fpi.row_gamma = gamma;
        fpi.row_a = a;
        fpi.row_nu = nu;
    } else if constexpr (METHOD == UVS_METHOD_IMCCKF) {          // K = c P H^T (c H P H^T + R)^-1 (experiment.py:262-264)
        kap = 1.0;
        gamma = c_shared * fast_rcp(fma(c_shared, a, 1.0));
    } else {                                                     // KF (experiment.py:192)
        kap = 1.0;
        gamma = fast_rcp(a + 1.0);
    }
    row_hook<4>(hook);
    if constexpr (!std::is_same<Share, NoShare>::value) {
        static_assert(METHOD == UVS_METHOD_KF || METHOD == UVS_METHOD_IMCCKF, "only estimators with one gain factor per filter share a block");
#pragma unroll
        for (int j = 0; j < N; ++j) share.g[j] = g[j];
        share.gamma = gamma;
    }
    // MCKF: an INFINITE innovation (a non-finite feature reached the filter) gives Cy = 0, inv(Cy) raises in the reference and the step keeps only
    // the prediction (gamma = 0 above) -- the state survives.  0 * inf must not turn it into NaN here: the innovation is clamped to the largest
    // finite value for the product (two instructions; a NaN innovation still arrives as a NaN gain and FAILs the trial as in the reference).
    const double step = (METHOD == UVS_METHOD_MCKF) ? gamma * fmax(fmin(nu, 1.7976931348623157e308), -1.7976931348623157e308) : gamma * nu;
    const double beta = gamma * (2.0 - gamma * (a + 1.0));
#pragma unroll
    for (int j = 0; j < N; ++j) {
        x[j] = fma(g[j], step, x[j]);                            // X + K (Z - H X) (experiment.py:291)
        chk = fma(x[j], 0.0, chk);
    }
    row_hook<5>(hook);
#pragma unroll
    for (int l = 0; l < N; ++l) {                                // Joseph update with R = 1: P -= beta g g^T
        const double w = beta * g[l];
#pragma unroll
        for (int j = l; j < N; ++j) pb[Sym<N>::at(l, j)] = fma(-w, g[j], pb[Sym<N>::at(l, j)]);
        if (l == 1) row_hook<6>(hook);
    }
}

template <int N, int METHOD, typename Hook>
UVS_DEV void rmckf_row(double (&x)[N], double (&pb)[Sym<N>::NP], const double (&dq)[N], double zi, double neg_half_inv_s2, double c_shared,
                       double reg, double &kap, double &chk, FpiProbe &fpi, Hook &hook) {
    NoShare none;
    rmckf_row<N, METHOD>(x, pb, dq, zi, neg_half_inv_s2, c_shared, reg, kap, chk, fpi, hook, none);
}

template <int N, int METHOD>
UVS_DEV void rmckf_row(double (&x)[N], double (&pb)[Sym<N>::NP], const double (&dq)[N], double zi, double neg_half_inv_s2, double c_shared,
                       double reg, double &kap, double &chk, FpiProbe &fpi) {
    NoHook none;
    rmckf_row<N, METHOD>(x, pb, dq, zi, neg_half_inv_s2, c_shared, reg, kap, chk, fpi, none);
}

// A row whose covariance block is the leading row's (see RowShare): innovation and state update only (experiment.py:274, 291).
template <int N>
UVS_DEV void rmckf_row_follow(double (&x)[N], const RowShare<N> &share, const double (&dq)[N], double zi, double &chk) {
    double pred = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) pred = fma(x[j], dq[j], pred);
    const double step = share.gamma * (zi - pred);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        x[j] = fma(share.g[j], step, x[j]);
        chk = fma(x[j], 0.0, chk);
    }
}

template <int N, int METHOD>
UVS_DEV void rmckf_row(double (&x)[N], double (&pb)[Sym<N>::NP], const double (&dq)[N], double zi, double neg_half_inv_s2, double c_shared,
                       double reg, double &kap, double &chk) {
    static_assert(METHOD != UVS_METHOD_MCKF, "MCKF rows need the fixed-point probe");
    FpiProbe unused;
    rmckf_row<N, METHOD>(x, pb, dq, zi, neg_half_inv_s2, c_shared, reg, kap, chk, unused);
}

// Householder QR least squares, rows interleaved over the L lanes of a filter: local row r of lane s is global row r*L + s.
// In column c the local row m = c / L is the pivot row on lane c % L, an ordinary "below" row on lanes > c % L and already
// finished on lanes < c % L; rows r > m are below the pivot on every lane -- so all lanes run the same unrolled code and only
// the treatment of row m is selected per lane.
// Returns true when the |R_cc| spread marks the Jacobian as numerically rank-deficient (rmckf_device.hpp, "numpy.linalg.pinv
// semantics"): the caller flags the trial and the careful second pass redoes it; the solution computed here is then discarded.
// nonfinite: some entry of the panel's Jacobian part is NaN or infinite (decided on NaN norms; see the end of the function for +inf).  A non-finite entry of column j reaches, through the reflector
// of an earlier column at the latest, every remaining row of column j, so the squared column norm n2 that column j's own step forms is
// non-finite: the exponent watch sees it for free, and the closed-loop kernel needs no separate finiteness probe of X (24 instructions per step).
template <int M, int N, int L>
UVS_DEV bool lstsq_tall_tuned(double (&a)[M / L][N + 1], int sub, double (&sol)[N], bool &nonfinite, bool certify = false) {
    constexpr int R = M / L;
    double rdiag[N];
    double rmax = 0.0;
    Spread spread;
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const int m = c / L, owner = c % L;
        const bool is_piv = (L == 1) || (sub == owner);
        const bool is_below = (L > 1) && (sub > owner);
        double sig = is_below ? a[m][c] * a[m][c] : 0.0;
#pragma unroll
        for (int r = m + 1; r < R; ++r) sig = fma(a[r][c], a[r][c], sig);
        sig = pair_sum<L>(sig);
        const double piv = pair_from_dyn<L>(a[m][c], owner);
        const double n2 = fma(piv, piv, sig);
        spread.add(n2);
        double nrm, rn;
        fast_sqrt_rsqrt_1(n2, nrm, rn);                                 // |R_cc| and its reciprocal
        // R_cc = -sign(piv) |column|; v_pivot = piv - R_cc = sign(piv) (|piv| + nrm): sign transfers (v_bfi), no compares or selects.
        // A column that vanished (n2 == 0) sends NaNs through the rest of the solve: nothing guards against it here, because such a trial is
        // marked (spread.lo == 0) and redone by the careful second pass whatever this solve returns.
        const double vp = piv + copysign(nrm, piv);
        // tau = 2 / (v.v) = 1 / (nrm (nrm + |piv|)) = rn / |vp|
        const double tau = rn * fast_rcp_1(fabs(vp));
        const double vm = is_piv ? vp : (is_below ? a[m][c] : 0.0);     // this lane's entry of the Householder vector in row m
#pragma unroll
        for (int j = c + 1; j <= N; ++j) {
            double d = vm * a[m][j];
#pragma unroll
            for (int r = m + 1; r < R; ++r) d = fma(a[r][c], a[r][j], d);
            d = pair_sum<L>(d) * tau;
            a[m][j] = fma(-d, vm, a[m][j]);
            // row c of R is final on its owner lane (the partner holds a row that is finished already, or one whose entries its column's
            // norm bounds): the running largest |R_cj|, one v_max_f64 with |.| modifiers per entry (see Spread::add_largest)
            if (j < N) rmax = fmax(rmax, fabs(a[m][j]));
#pragma unroll
            for (int r = m + 1; r < R; ++r) a[r][j] = fma(-d, a[r][c], a[r][j]);
        }
        rdiag[c] = -copysign(rn, piv);                                  // 1 / R_cc straight from the rsqrt
    }
    if constexpr (L > 1) rmax = fmax(rmax, dpp_quad<kSwapPair>(rmax));
    if constexpr (L == 4) rmax = fmax(rmax, dpp_quad<kSwapHalf>(rmax));
    spread.add_largest(rmax);
#pragma unroll
    for (int c = N - 1; c >= 0; --c) {
        const int m = c / L, owner = c % L;
        double rhs = a[m][N];
#pragma unroll
        for (int j = c + 1; j < N; ++j) rhs = fma(-a[m][j], sol[j], rhs);
        rhs = pair_from_dyn<L>(rhs, owner);
        sol[c] = rhs * rdiag[c];
    }
    // solution growth (Spread::grows, round 5): the largest solution entry against the largest of the top N entries of Q^T y
    double smax = fabs(sol[0]), cmax = 0.0;
#pragma unroll
    for (int c = 1; c < N; ++c) smax = fmax(smax, fabs(sol[c]));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r * L < N) cmax = fmax(cmax, ((r + 1) * L <= N || r * L + sub < N) ? fabs(a[r][N]) : 0.0);
    }
    // (no exchange: a lane's verdict rests on ITS rows of Q^T y -- on lane 0 of the filter, whose mark is the one that is written, global rows
    // 0, L, 2 L, ...; a smaller denominator only marks sooner, and the healthy fixtures stay five orders of magnitude below the gate)
    const bool grows = spread.grows(smax, cmax);
    // (a column that vanished exactly -- lo == 0 -- sends NaNs through the remaining columns by itself: that trial is marked for the careful
    // second pass, which probes X entry by entry, and is not FAILed here)
    // NaN (high dword above +inf's 0x7ff00000) proves a non-finite entry.  A norm of exactly +inf does not: a FINITE entry beyond ~1e154
    // overflows the square, and numpy's SVD does not raise on that -- such a trial is marked for the careful pass (which probes X entry by
    // entry) instead of FAILed here.  An infinite entry in any but the last column turns a later column's norm into NaN (0 * inf in the
    // reflector); in the last column it goes the careful way too and FAILs there, at the same step.
    nonfinite = spread.hi > 0x7ff00000u && spread.lo != 0u;
    bool uncertified = false;
#ifdef UVS_NO_CERTIFICATE               // diagnostic build: A/B of what the cold strict-mode branch costs the plain step (register allocation)
    certify = false;
#endif
    if (__builtin_expect(certify, 0)) {
        // UVS_OPT_STRICT_PINV (round 6): a CERTIFICATE instead of a heuristic.  numpy's pinv (experiment.py:312) drops singular values below
        // 1e-15 sigma_max; when none is that small, pinv(J) y IS the least-squares solution just computed.  cond_2(R) <= |R|_F |R^-1|_F, so the
        // inverse of the triangle, column by column over the rows where they live (the back substitution above, six times, unit right-hand sides),
        // bounds the condition number from ABOVE: below ~2^42 the solve is certified -- a margin of 2^7 to numpy's cutoff for what rounding does to
        // the computed inverse at that conditioning -- and anything else marks the trial for the SVD pass, which then decides by the singular
        // values themselves.  |R|_F^2 <= 21 max^2 comes from the spread's running maximum.  ~170 instructions per step, in strict mode only.
        double inv2 = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double z[N];
            z[k] = rdiag[k];
            inv2 = fma(z[k], z[k], inv2);
#pragma unroll
            for (int c = k - 1; c >= 0; --c) {
                const int m = c / L, owner = c % L;
                double acc = 0.0;
#pragma unroll
                for (int j = c + 1; j <= k; ++j) acc = fma(a[m][j], z[j], acc);
                z[c] = -pair_from_dyn<L>(acc, owner) * rdiag[c];
                inv2 = fma(z[c], z[c], inv2);
            }
        }
        // high dwords add like exponents: certified when max^2 |R^-1|_F^2 < 2^78 up to the fields' slack (a factor 4), i.e. -- with 21 entries in
        // |R|_F^2 -- when cond^2 < 21 * 2^80 < 2^85
        const unsigned long long lhs = (unsigned long long)(unsigned)__double2hiint(inv2) + spread.hi;
        uncertified = !(lhs < 2ull * 0x3ff00000u + (78ull << 20));          // (NaN / inf in either factor compare as "not below")
    }
    return spread.suspect() || spread.hi == 0x7ff00000u || grows || uncertified;
}

template <int M, int N, int L>
UVS_DEV bool lstsq_tall_tuned(double (&a)[M / L][N + 1], int sub, double (&sol)[N]) {
    bool unused;
    return lstsq_tall_tuned<M, N, L>(a, sub, sol, unused);
}

// ------------------------------------------------------------------------------------------------ four lanes, the two-lane kernel's bits
// EMU2: a filter on the four lanes of a quad that reproduces the TWO-lane kernel's arithmetic bit for bit, so that the launcher may pick it for
// batches that do not fill the chip without changing a single result (an N-GPU sweep returns the bits of the 1-GPU sweep).  Quad lane
// `sub` = p + 2 h: p is the parity the two-lane kernel's lane has (u rows / v rows, kinematic half chain), h says which half of that lane's
// four local rows this lane holds (two-lane local row R2 = 2 h + r, global row 2 R2 + p).  Everything lane-local is the two-lane code on half
// the rows; every sum the two-lane kernel forms as a sequential chain over its four local rows is formed here in the same order -- h = 0 starts
// it, hands it to h = 1 (quad_perm [0,1,0,1]), which finishes it; the pair sum across p and a broadcast back (quad_perm [2,3,2,3]) follow.
constexpr int kQuadFromLow = 0x44;       // quad_perm [0,1,0,1]: value of the h = 0 lane of the same parity
constexpr int kQuadFromHigh = 0xEE;      // quad_perm [2,3,2,3]: value of the h = 1 lane of the same parity
// total = (chain finished on the h = 1 lanes) summed over the two parities, delivered to all four lanes
UVS_DEV double emu2_finish(double chain_on_high) {
    // both parities' finished chains fetched independently (quad_perm [2,2,2,2] and [3,3,3,3]) and added: the two-lane kernel's own + partner's,
    // commutative, so every lane holds its bits -- one cross-lane hop on the critical path instead of two (add on the h = 1 lanes, then broadcast)
    return dpp_quad<0xAA>(chain_on_high) + dpp_quad<0xFF>(chain_on_high);
}
// Householder least squares of the 8 x (6 + 1) panel: a[r][.] is the lane's local row r (two-lane local row 2 h + r).  Same operations on the
// same values in the same order as lstsq_tall_tuned<8, 6, 2>; see there for the algorithm and for what `nonfinite` and the return value mean.
template <int M, int N>
UVS_DEV bool lstsq_tall_emu2(double (&a)[2][N + 1], int sub, double (&sol)[N], bool &nonfinite) {
    static_assert(M == 8, "EMU2 splits the four local rows of a two-lane filter over two lanes");
    const int p = sub & 1;
    const bool high = sub & 2;
    double rdiag[N];
    double rmax_lo = 0.0, rmax_hi = 0.0;                            // largest |R_cj| of the rows finished on the h = 0 / h = 1 lanes
    Spread spread;
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const int m = c / 2, owner = c % 2;                         // two-lane local row of the pivot, parity that owns it
        const int hm = m / 2, rm = m % 2;                           // ... which lives on the h = hm lanes at local row rm
        const bool is_piv = (p == owner), is_below = (p > owner);
        // squared norm of the column below the pivot: two-lane order = (row m if below the pivot), rows m + 1 .. 3 of the lane, then the other parity
        double sig;
        {
            const double seed = is_below ? a[rm][c] * a[rm][c] : 0.0;
            if (hm == 0) {
                double s0 = seed;
                if (rm == 0) s0 = fma(a[1][c], a[1][c], s0);
                double s1 = dpp_quad<kQuadFromLow>(s0);
                s1 = fma(a[0][c], a[0][c], s1);
                s1 = fma(a[1][c], a[1][c], s1);
                sig = emu2_finish(s1);
            } else {
                double s1 = seed;
                if (rm == 0) s1 = fma(a[1][c], a[1][c], s1);
                sig = emu2_finish(s1);
            }
        }
        const double piv = pair_from_dyn<4>(a[rm][c], owner + 2 * hm);
        const double n2 = fma(piv, piv, sig);
        spread.add(n2);
        double nrm, rn;
        fast_sqrt_rsqrt_1(n2, nrm, rn);
        const double vp = piv + copysign(nrm, piv);
        const double tau = rn * fast_rcp_1(fabs(vp));
        const double vm = is_piv ? vp : (is_below ? a[rm][c] : 0.0);     // entry of the Householder vector in two-lane local row m
        // this lane's entries of the vector in its local rows 0, 1: the h = hm lanes hold row m (and, below it, ordinary rows); the h = 1 lanes hold only
        // ordinary rows while hm = 0; the h = 0 lanes are finished once hm = 1 (zero: they neither contribute nor change)
        double v[2];
        if (hm == 0) {
            v[0] = high ? a[0][c] : (rm == 0 ? vm : 0.0);
            v[1] = high ? a[1][c] : (rm == 1 ? vm : a[1][c]);
        } else {
            v[0] = high ? (rm == 0 ? vm : 0.0) : 0.0;
            v[1] = high ? (rm == 1 ? vm : a[1][c]) : 0.0;
        }
#pragma unroll
        for (int j = c + 1; j <= N; ++j) {
            double d1;
            if (hm == 0) {
                double d0 = v[rm] * a[rm][j];
                if (rm == 0) d0 = fma(v[1], a[1][j], d0);
                d1 = dpp_quad<kQuadFromLow>(d0);
                d1 = fma(v[0], a[0][j], d1);
                d1 = fma(v[1], a[1][j], d1);
            } else {
                d1 = v[rm] * a[rm][j];
                if (rm == 0) d1 = fma(v[1], a[1][j], d1);
            }
            const double d = emu2_finish(d1) * tau;
            // rows that are finished stay untouched, as in the two-lane code (a product with a zero entry could still flip the sign of a zero)
            const double u0 = fma(-d, v[0], a[0][j]), u1 = fma(-d, v[1], a[1][j]);
            if (hm == 0) {
                a[0][j] = (rm == 1) ? (high ? u0 : a[0][j]) : u0;
                a[1][j] = u1;
            } else {
                a[0][j] = (rm == 0) ? (high ? u0 : a[0][j]) : a[0][j];
                a[1][j] = high ? u1 : a[1][j];
            }
            if (j < N) { if (hm == 0) rmax_lo = fmax(rmax_lo, fabs(a[rm][j])); else rmax_hi = fmax(rmax_hi, fabs(a[rm][j])); }
        }
        rdiag[c] = -copysign(rn, piv);
    }
    {   // the two-lane kernel's watch: rows 2 m + {0, 1} for every column -- here the h = hm lanes of both parities
        double rmax = fmax(dpp_quad<kQuadFromLow>(rmax_lo), dpp_quad<kQuadFromHigh>(rmax_hi));
        rmax = fmax(rmax, dpp_quad<kSwapPair>(rmax));
        spread.add_largest(rmax);
    }
#pragma unroll
    for (int c = N - 1; c >= 0; --c) {
        const int m = c / 2, owner = c % 2, hm = m / 2, rm = m % 2;
        double rhs = a[rm][N];
#pragma unroll
        for (int j = c + 1; j < N; ++j) rhs = fma(-a[rm][j], sol[j], rhs);
        rhs = pair_from_dyn<4>(rhs, owner + 2 * hm);
        sol[c] = rhs * rdiag[c];
    }
    // solution growth: the two-lane kernel's verdict -- the same maxima on the lane whose mark is written (global rows < N of its parity: both rows of the h = 0 lane, row 0 of the h = 1 lane)
    double smax = fabs(sol[0]), cmax = fabs(a[0][N]);
#pragma unroll
    for (int c = 1; c < N; ++c) smax = fmax(smax, fabs(sol[c]));
    cmax = fmax(cmax, high ? 0.0 : fabs(a[1][N]));
    cmax = fmax(cmax, dpp_quad<kSwapHalf>(cmax));                   // lane 0 of the quad: global rows 0, 2 (its own) and 4 -- the rows of the two-lane kernel's lane 0
    nonfinite = spread.hi > 0x7ff00000u && spread.lo != 0u;
    return spread.suspect() || spread.hi == 0x7ff00000u || spread.grows(smax, cmax);
}

// Internal plant kind (not part of the ABI): UVS_PLANT_DH_PINHOLE whose DH table has, in either half of a six-link chain, alpha = -pi/2 on
// the first link and alpha = 0 on the last (the UR10: -pi/2, 0, 0 | -pi/2, pi/2, 0).  The launcher selects it when the table says so.
constexpr int kPlantDhAxisAligned = 2;

// Plant constants are broadcast from LDS (one ds_read per pair of doubles) instead of sitting in ~90 SGPRs that the register
// allocator would spill to VGPR lanes and fetch back with v_readlane + s_nop on every use.
template <int M, int N>
struct PlantLds {
    static constexpr int kJoint = 0;                   // 5 doubles per joint: theta_offset, d, a, cos_alpha, sin_alpha
    static constexpr int kPoint = 5 * N;               // 3 doubles per point
    static constexpr int kCam = kPoint + 3 * (M / 2);  // focal, center
    static constexpr int kCount = kCam + 2;
};

// PV = number of this lane's covariance blocks kept in VGPRs; the other R - PV blocks live in LDS and pass through registers
// only while their row is updated (an LDS round trip moves two doubles per instruction, an AGPR one half a double).
// XOUT = write the per-step X stream.
//
// Store discipline: the step body is straight-line code.  Lanes of a padding trial shadow the last real trial (same inputs,
// same values, same addresses: their stores are duplicates), both lanes of a pair store the replicated q / dq, and a trial
// that FAILs keeps running on NaNs -- rows at and after its k_done are unspecified, exactly the rows the reference trims
// (experiment.py:345-352).
// With L = 4 the whole state fits the 256 VALU-addressable registers (P: 84, X: 24), LDS holds only the statistics and
// the plant constants, and two wavefronts share a SIMD (XREG = true, __launch_bounds__(64, 2)): measured 1.36x the fp64
// issue rate of a lone wavefront, with scalar/LDS/memory instructions of one wavefront hidden under the other's arithmetic.
constexpr int kFairPrioLog2 = 18;       // log2 of the priority turn in shader clocks for the two-wavefront-per-SIMD kernels (12 / 15 / 18 measured, appendix A.2)
// Wavefronts per SIMD the four-lane kernels are compiled for.  Round 3 shipped 2 (256 registers: the RMCKF instantiation then carried 36-68 B of
// scratch inside the step loop); at 1 it takes 270 registers, no scratch, and is faster at every size measured (profiles/r04/shard_times.txt:
// 8 192 trials 0.97 -> 0.90 ms, 16 384: 1.26 -> 1.00, 32 768: 2.06 -> 1.88).  Four lanes per filter are the LATENCY mapping (UVS_OPT_LATENCY):
// half the trials per wavefront, 14 % fewer instructions per wavefront-step -- the shards of a strong-scaling series that do not fill the chip.
constexpr int kL4Occ = 1;
constexpr int kSharedOcc = 2;           // wavefronts per SIMD of the two-lane KF / IMCC-KF kernels (one covariance block per lane)
// SEGMENTED: the instantiation can run a trial chunk as several work items (see SEG below).  Always for MCKF, whose wavefronts differ in length;
// for RMCKF a second instantiation that the launcher picks only when a launch is not a whole number of rounds of wavefronts (the code costs the
// headline kernel 4 registers and 0.4 %, so the headline launch keeps the instantiation without it).
// XREC (round 5): the X stream as per-trial RECORDS -- [step][trial][m n], comp_stride 1 -- written straight out of the LDS-resident X after the
// rows of a step: 12 stores of 16 bytes per lane, each covering eight whole 128-byte lines, instead of 24 stores of 8 bytes per lane scattered
// over 48 rows of the trial-fastest layout.  KF / IMCC-KF are bound by the CU's store path; the launcher picks this
// instantiation when the caller's x_out view has that shape.
// CERT (round 6): UVS_OPT_STRICT_PINV's certificate (lstsq_tall_tuned) is a uniform run-time branch in the RMCKF kernels, which it costs nothing
// (same registers, same main path).  In the MCKF kernel even the untaken branch cost the plain step 21 instructions of register shuffling (+ 0.8 %,
// profiles/r06/strict_certificate_ab.txt), and the KF / IMCC-KF kernels, held to 256 registers for two wavefronts per SIMD, spilled 12 bytes over
// it: those three take it as a compile-time switch, and the launcher picks their CERT instantiations in strict mode.
// XPAIR (round 6; with XREC): the X stream leaves LDS as 16-byte pairs of consecutive trials into TRIAL-FASTEST rows instead of as records (store_records).
template <int M, int N, int L, int METHOD, int PLANT, int PV, bool XOUT, bool EMU2 = false, bool SEGMENTED = (METHOD == UVS_METHOD_MCKF), bool XREC = false,
          bool CERT = false, bool XPAIR = false>
__global__ __launch_bounds__(64, (L >= 4 ? kL4Occ : ((METHOD == UVS_METHOD_KF || METHOD == UVS_METHOD_IMCCKF) && PV >= 1 && L == 2) ? kSharedOcc : 1))
void closed_loop_tuned_kernel(const ClosedArgs A) {
    static_assert(M >= N && (L == 1 || L == 2 || L == 4) && M % L == 0, "tuned kernel: tall Jacobian, 1, 2 or 4 lanes per filter");
    static_assert(!XREC || (XOUT && L == 2 && !EMU2 && M == 8 && N == 6 && METHOD != UVS_METHOD_MCKF), "record stores: the (8,6) two-lane kernels with X in LDS (MCKF rewrites rows of a step)");
    static_assert(!XPAIR || XREC, "pair stores are a flavour of the LDS store path");
    constexpr bool XREG = (L >= 4);                                // X in registers instead of LDS
    // MCKF trials differ in length (a trial that iterates costs its whole wavefront the fixed-point branch): with exactly two rounds of
    // wavefronts a slow one serialises with its slot's second wavefront.  The two-lane MCKF kernel can therefore run a chunk's K steps as
    // A.n_seg work items, the state crossing through HBM (uvs_rmckf_closed_loop_ws_f64); same arithmetic, bit-identical results.  The RMCKF
    // kernel has a SEGMENTED instantiation for launches that are not a whole number of rounds (1.5 rounds take two rounds' time otherwise).
    constexpr bool SEG = (SEGMENTED && L == 2 && !XREG && PLANT != UVS_PLANT_LINEAR);   // (the linear-plant instantiations have no registers to spare for it)
    // Split kinematics: the L lanes of a filter form G groups, group g multiplies links g*JG .. g*JG+JG-1 of the DH chain and
    // keeps only those joints' angles; the camera pose is assembled from the G partial products with DPP broadcasts.
    constexpr bool DH = (PLANT == UVS_PLANT_DH_PINHOLE || PLANT == kPlantDhAxisAligned);
    constexpr bool AXIS = (PLANT == kPlantDhAxisAligned);
    static_assert(!EMU2 || (L == 4 && M == 8 && N == 6), "EMU2: the (8,6) two-lane arithmetic on the four lanes of a quad");
    constexpr bool HALVES = (L == 2 || EMU2);                      // the kinematic chain in two halves of three links (else three groups of two)
    static_assert(!AXIS || (HALVES && N == 6), "the axis-aligned chain is written for three links per lane");
    constexpr bool SPLIT = DH && (L == 2 || L == 4) && (N % (HALVES ? 2 : 3) == 0);
    constexpr int G = SPLIT ? (HALVES ? 2 : 3) : 1;
    constexpr int JG = N / G;                                      // joints tracked by this lane
    constexpr int R = M / L, NP = Sym<N>::NP, TPW = 64 / L;       // rows per lane, packed block size, trials per wavefront
    static_assert(PV >= 0 && PV <= R, "PV counts covariance blocks");
    constexpr int PL = R - PV;                                     // blocks resident in LDS
    constexpr bool SHARED_P = (METHOD == UVS_METHOD_KF || METHOD == UVS_METHOD_IMCCKF) && PV >= 1;   // identical blocks: keep one (RowShare)
    using PC = PlantLds<M, N>;
    __shared__ double lds_x[XREG ? 1 : R * N][64];
    __shared__ double lds_acc[3 * R][64];
    __shared__ double lds_p[(PL > 0 && !((METHOD == UVS_METHOD_KF || METHOD == UVS_METHOD_IMCCKF) && PV >= 1)) ? PL * NP : 1][64];
    __shared__ double lds_c[PC::kCount];

    // diagnostics (dead code unless a kDiag* switch is on): entry time of the wavefront / work item; entry -> state ready -> steps done -> handed over
    unsigned long long wt_first = 0, it_t0 = 0, it_t1 = 0, it_t2 = 0, it_t3 = 0, it_ta = 0, it_tb = 0;
    if constexpr (kDiagWaves) wt_first = diag_ticks();
    if constexpr (kDiagItems) it_t0 = diag_ticks();
    const unsigned lane = threadIdx.x;
    const int sub = (L == 1) ? 0 : (int)(lane & (L - 1));
    const int grp = SPLIT ? (EMU2 ? (sub & 1) : (sub < G ? sub : G - 1)) : 0;   // with L = 4 the fourth lane mirrors group 2 (EMU2: group = parity)
    // rows of a lane: local row r is global row r * RS + rb (interleaved over the L lanes; EMU2: 4 h + 2 r + p, the two-lane kernel's rows 2 (2 h + r) + p)
    constexpr int RS = EMU2 ? 2 : L;
    const int rb = EMU2 ? 4 * (sub >> 1) + (sub & 1) : sub;
    auto own_row = [&](auto &&at, int r) {                         // at(global row) for this lane's local row r, without a variably indexed access
        double cand[L];
#pragma unroll
        for (int u = 0; u < L; ++u) cand[u] = at(r * RS + (EMU2 ? 4 * (u >> 1) + (u & 1) : u));
        return pick_sub<L>(cand, sub);
    };
    // Work item = (trial chunk, segment).  Workgroups are dispatched in the order of their ids, so with ids laid out segment-major every
    // item's predecessor (same chunk, previous segment: an id smaller by the number of chunks) was dispatched before it and runs to its
    // end without waiting on anything later -- the hardware dispatcher is the work queue (MCKF).
    long long chunk = blockIdx.x;
    int seg = 0;
    if constexpr (SEG) {
        if (A.n_seg > 1) {
            const unsigned nchunks = gridDim.x / (unsigned)A.n_seg;
            seg = (int)(blockIdx.x / nchunks);
            chunk = blockIdx.x - (unsigned)seg * nchunks;
        }
    }
    const long long wave_first = chunk * TPW;                      // first trial of this wavefront (uniform)
    const unsigned tl = lane / L;                                   // trial within the wavefront
    const bool valid = wave_first + tl < A.T;
    const long long trial = valid ? wave_first + tl : A.T - 1;     // padding lanes shadow the last trial
    const uvs_filter_params &fp = A.fp;
    const int K = fp.steps;
    // segment [k_begin, k_end) of the trial; `fresh`: start from the initial state (first segment, or a later one whose predecessor did not
    // report within the spin budget -- it then recomputes the trial from step 0, writing the same rows once more: never a deadlock)
    int k_begin = 0, k_end = K, flag_early = 0;
    bool fresh = true, last_seg = true;
    if constexpr (SEG) {
        if (A.n_seg > 1) {
            k_begin = A.seg_first[seg];
            k_end = A.seg_first[seg + 1];
            last_seg = seg == A.n_seg - 1;
            // the predecessor's counter is asked for NOW (relaxed) and looked at after the part of the prologue that does not depend on it --
            // cursors, plant constants, the first noise rows: an item's start was three memory round trips one after the other (round 5:
            // 12-15 us from entry to the first step, tools/read_stamps.py --items)
            if (seg > 0) flag_early = __hip_atomic_load(&A.ws_flags[chunk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    auto await_predecessor = [&]() {
        if constexpr (SEG) {
            if (A.n_seg > 1 && seg > 0) {
                bool there = flag_early >= seg;
                if (there) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // the state loads below must not be served from lines cached before the counter moved
                } else {
                    int spins = 0;
                    do {
                        there = __hip_atomic_load(&A.ws_flags[chunk], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= seg;
                        if (there) break;
                        __builtin_amdgcn_s_sleep(32);
                    } while (++spins < kSegSpinMax);
                }
                there = __builtin_amdgcn_readfirstlane((int)there) != 0;          // (every lane ran the loads itself)
                fresh = !there;
                if (fresh) {                                                      // fallback: recompute from step 0 -- and say so: the word behind the chunks'
                    k_begin = 0;                                                  // counters counts the items that ran out their budget (0 on a healthy launch)
                    if (lane == 0) atomicAdd(&A.ws_flags[gridDim.x / (unsigned)A.n_seg], 1);
                }
            }
        }
    };

    // per-lane stream cursors (advance by the step stride once per step; components are reached by adding the uniform stride)
    const double *pn = A.noise.p ? A.noise.at(trial, 0, rb) : nullptr;
    double *px = A.x_out.p ? A.x_out.at(trial, 0, rb * N) : nullptr;
    double *pe = A.err_out.p ? A.err_out.at(trial, 0, rb) : nullptr;
    double *pf = A.f_out.p ? A.f_out.at(trial, 0, rb) : nullptr;
    double *pq = A.q_out.p ? A.q_out.at(trial, 0, grp * JG) : nullptr;
    double *pd = A.dq_out.p ? A.dq_out.at(trial, 0, grp * JG) : nullptr;
    // Component addresses are formed from the step cursor with the whole index expression (cursor + (row, column) * comp_stride), not by
    // walking a second pointer: one live 64-bit value per stream instead of two -- which is what kept the MCKF instantiation out of scratch.
    const bool on_noise = A.noise.p != nullptr, on_err = A.err_out.p != nullptr, on_f = A.f_out.p != nullptr,
               on_q = A.q_out.p != nullptr, on_dq = A.dq_out.p != nullptr;

    double q[JG], dq[N], f_prev[R], des[R];                        // q: this lane's joints; dq: replicated command
    double sn[JG], cs[JG];                                         // sin / cos of this lane's joint angles, carried from step to step
    bool reseed = true;
#pragma unroll
    for (int u = 0; u < JG; ++u) { sn[u] = 0.0; cs[u] = 1.0; }
    double p[PV > 0 ? PV : 1][NP];
    double xr[XREG ? R : 1][N];                                    // X when it lives in registers
    double t = fp.dt;
    int status = UVS_STATUS_SUCCESS, k_done = K;
    bool alive = true, flagged = false;                          // flagged: a rank-deficient Jacobian was seen -> careful second pass
#pragma unroll
    for (int r = 0; r < R; ++r) des[r] = own_row([&](int row) { return fp.desired[row]; }, r);
    double restored_word = 0.0;
    // The state of a trial chunk between two of its segments, [field][lane] in the workspace (seg_state_doubles per lane).
    auto seg_state = [&](auto saving) {
        constexpr bool SAVE = decltype(saving)::value;
        if constexpr (SEG) {
            double *w = A.ws_state + chunk * (long long)(seg_state_doubles(M, N, L) * 64) + lane;
            // Saving: agent-scope stores (write through this XCD's L2).  Plain stores would need a release fence to become visible to the
            // XCD that runs the next segment, and on gfx950 that fence is `buffer_wbl2`: a write-back of the whole L2, 4 MB of other
            // wavefronts' streaming output -- measured at ~55 us per hand-over.
            auto put = [&](double *at, double v) { __hip_atomic_store(at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            auto io = [&](double &v) { if constexpr (SAVE) put(w, v); else v = *w; w += 64; };
#pragma unroll
            for (int u = 0; u < JG; ++u) { io(q[u]); io(sn[u]); io(cs[u]); }
#pragma unroll
            for (int j = 0; j < N; ++j) io(dq[j]);
#pragma unroll
            for (int r = 0; r < R; ++r) io(f_prev[r]);
            io(t);
#pragma unroll
            for (int r = 0; r < PV; ++r)
#pragma unroll
                for (int e = 0; e < NP; ++e) io(p[r][e]);
            // the LDS-resident part in batches: all loads of a batch in flight before the first is consumed (one at a time, each restore would
            // wait out a full memory latency per value)
            constexpr int NLDS = PL * NP + R * N + 3 * R;
            auto lds_cell = [&](int i) -> double & {
                return i < PL * NP ? lds_p[i < PL * NP ? i : 0][lane] : (i < PL * NP + R * N ? lds_x[(i - PL * NP) % (R * N)][lane] : lds_acc[(i - PL * NP - R * N) % (3 * R)][lane]);
            };
            // Restoring (round 5): [field][lane] of the workspace IS the LDS image of these arrays (a row of 64 doubles = 512 bytes), so the
            // LDS-resident part is copied by `global_load_lds_dwordx4` -- lane l moves bytes [16 l, 16 l + 16) of every KB, no VGPRs and no
            // batches: 39 requests in flight behind one another instead of two round trips of 52 / 26 (measured: restore 6.8-8.8 us per item
            // before, tools/read_stamps.py --items)
            constexpr bool DMA = !SAVE && !XREG && (PL * NP) % 2 == 0 && (R * N) % 2 == 0 && (3 * R) % 2 == 0 && !SHARED_P && PL > 0;
            if constexpr (DMA) {
                typedef __attribute__((address_space(1))) const void gvoid_t;
                typedef __attribute__((address_space(3))) void lvoid_t;
                const char *g = reinterpret_cast<const char *>(w - lane) + 16 * lane;
                auto rows = [&](double *first, int count) {              // `count` rows of 64 doubles
#pragma unroll
                    for (int j = 0; j < count / 2; ++j)
                        __builtin_amdgcn_global_load_lds((gvoid_t *)(g + 1024 * j), (lvoid_t *)(reinterpret_cast<char *>(first) + 1024 * j), 16, 0, 0);
                    g += 512 * count;
                };
                rows(&lds_p[0][0], PL * NP);
                rows(&lds_x[0][0], R * N);
                rows(&lds_acc[0][0], 3 * R);
                w += NLDS * 64;
            }
            constexpr int BATCH = 52;                                    // (vmcnt lets 63 of them be in flight at once; 52 = two batches at (8,6), and the size at which the MCKF instantiation spills least)
#pragma unroll
            for (int b = 0; b < (DMA ? 0 : NLDS); b += BATCH) {
                double tmp[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i)
                    if (b + i < NLDS) tmp[i] = SAVE ? lds_cell(b + i) : w[(long long)i * 64];
#pragma unroll
                for (int i = 0; i < BATCH; ++i)
                    if (b + i < NLDS) { if constexpr (SAVE) put(w + (long long)i * 64, tmp[i]); else lds_cell(b + i) = tmp[i]; }
                w += (NLDS - b < BATCH ? NLDS - b : BATCH) * 64;
            }
            const int bits = (alive ? 1 : 0) | (flagged ? 2 : 0) | (reseed ? 4 : 0) | (status << 8);
            double word = __hiloint2double(k_done, bits);
            io(word);
            if constexpr (!SAVE) restored_word = word;                 // decoded by the caller, AFTER the rest of the prologue: consuming it here would wait out every load above
        }
    };
    await_predecessor();
    if constexpr (kDiagItems) it_ta = diag_ticks();
    if (!fresh) {
        seg_state(std::false_type{});
        if constexpr (kDiagItems) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); it_tb = diag_ticks(); }
    } else {
        double q_all[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { q_all[j] = *A.q_start.at(trial, 0, j); dq[j] = 0.0; }
#pragma unroll
        for (int u = 0; u < JG; ++u) {
            double cand[G];
#pragma unroll
            for (int g = 0; g < G; ++g) cand[g] = q_all[g * JG + u];
            q[u] = (G == 1) ? cand[0] : (G == 2 ? (grp ? in_reg(cand[1]) : in_reg(cand[0])) : (grp == 0 ? in_reg(cand[0]) : (grp == 1 ? in_reg(cand[1]) : in_reg(cand[G - 1]))));
        }
        double x0[R][N];
        if (fp.initial_guess) {
            double xa[M][N], fa[M];                                // all rows, then keep this lane's interleaved share
            initial_guess<M, N, 1>(A.plant, q_all, 0, xa, fa);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f_prev[r] = own_row([&](int row) { return fa[row]; }, r);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    x0[r][j] = own_row([&](int row) { return xa[row][j]; }, r);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f_prev[r] = 0.0;                                   // f = zeros(m) (experiment.py:56)
#pragma unroll
                for (int j = 0; j < N; ++j) x0[r][j] = *A.x0.at(trial, 0, (r * RS + rb) * N + j);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if constexpr (XREG) xr[r][j] = x0[r][j];
                else lds_x[r * N + j][lane] = x0[r][j];
            }
        }
#pragma unroll
        for (int i = 0; i < 3 * R; ++i) lds_acc[i][lane] = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = l; j < N; ++j) {
                    const double v = (l == j) ? 1.0 : 0.0;                     // P = I (experiment.py:73)
                    if (r < PV) p[r < PV ? r : 0][Sym<N>::at(l, j)] = v;
                    else if constexpr (!SHARED_P) lds_p[(r - PV) * NP + Sym<N>::at(l, j)][lane] = v;
                }
    }
    // (round 6) The rest of the prologue runs UNDER the state restore: a later work item issues its 63 register loads and 39 LDS-direct requests
    // first (5-6 us to arrive), then fills the plant constants and asks for its first noise rows; the flags word is decoded after that.
    double nz_early[R];                                            // the first noise rows of the segment as planned (a fallback reloads those of step 0)
    const int k_planned = k_begin;
#pragma unroll
    for (int r = 0; r < R; ++r) nz_early[r] = 0.0;
    if (on_noise && k_planned < K) {
#pragma unroll
        for (int r = 0; r < R; ++r) nz_early[r] = (pn + (long long)k_planned * A.noise.sk)[r * RS * A.noise.sc];
    }
    if constexpr (DH) {
        if (lane < N) {
            lds_c[PC::kJoint + 5 * lane + 0] = A.plant.theta_offset[lane];
            lds_c[PC::kJoint + 5 * lane + 1] = A.plant.d[lane];
            lds_c[PC::kJoint + 5 * lane + 2] = A.plant.a[lane];
            lds_c[PC::kJoint + 5 * lane + 3] = A.plant.cos_alpha[lane];
            lds_c[PC::kJoint + 5 * lane + 4] = A.plant.sin_alpha[lane];
        }
        if (lane < M / 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) lds_c[PC::kPoint + 3 * lane + c] = A.plant.points[lane][c];
        }
        if (lane == 0) { lds_c[PC::kCam] = A.plant.focal; lds_c[PC::kCam + 1] = A.plant.center; }
    }

    if constexpr (SEG) {
        if (!fresh) {
            const int b = __double2loint(restored_word);
            alive = b & 1; flagged = b & 2; reseed = b & 4; status = b >> 8;
            k_done = __double2hiint(restored_word);
        }
    }
    __syncthreads();                                               // lds_c is read by every lane

    double nz_next[R];
#pragma unroll
    for (int r = 0; r < R; ++r) nz_next[r] = 0.0;
    if constexpr (SEG) {
        if (k_begin > 0) {                                       // a later segment: every stream cursor to its first step
            if (on_noise) pn += (long long)k_begin * A.noise.sk;
            if constexpr (XOUT && !XREC) px += (long long)k_begin * A.x_out.sk;
            if (on_err) pe += (long long)k_begin * A.err_out.sk;
            if (on_f) pf += (long long)k_begin * A.f_out.sk;
            if (on_q) pq += (long long)k_begin * A.q_out.sk;
            if (on_dq) pd += (long long)k_begin * A.dq_out.sk;
        }
        if (!__any(alive)) k_end = k_begin;                      // every trial of the chunk FAILed in an earlier segment
    }
    if (on_noise && k_begin < K) {
        if (k_begin == k_planned) {
#pragma unroll
            for (int r = 0; r < R; ++r) nz_next[r] = nz_early[r];
        } else {                                                 // (the fallback: this item starts over at step 0)
#pragma unroll
            for (int r = 0; r < R; ++r) nz_next[r] = pn[r * RS * A.noise.sc];
        }
        pn += A.noise.sk;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): keep "nz_next may be in flight" out of the loop header (see rmckf_replay_tuned.hpp)
    if constexpr (kDiagItems) it_t1 = diag_ticks();
    DiagPhases diag_steps, diag_fpi;
    unsigned long long rt_first = 0, fpi_loop0 = 0;              // slot 4 of diag_steps = wall ticks of the loop (-> shader clock)
    if constexpr (kDiagStamps) { diag_steps.last = diag_cycles(); rt_first = diag_ticks(); }
    auto record_wave = [&]() {                                   // (kDiagWaves) this wavefront's interval and place into its slice of `stats`
        if (lane == 0 && A.stats) {
            unsigned hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
            double *w_ = A.stats + 3 * wave_first + 4 * seg;
            w_[0] = (double)wt_first; w_[1] = (double)diag_ticks(); w_[2] = (double)hw_id; w_[3] = (double)xcc_id;
        }
    };
    auto record_item = [&](bool handed_over) {                   // (kDiagItems) sums per segment over the first words of `stats`
        if (lane == 0 && A.stats) {
            atomicAdd(&A.stats[8 * seg + 0], (double)(it_t1 - it_t0));
            atomicAdd(&A.stats[8 * seg + 1], (double)(it_t2 - it_t1));
            if (handed_over) atomicAdd(&A.stats[8 * seg + 2], (double)(it_t3 - it_t2));
            atomicAdd(&A.stats[8 * seg + 3], 1.0);
            atomicAdd(&A.stats[8 * seg + 4], (double)(it_ta - it_t0));
            if (it_tb) atomicAdd(&A.stats[8 * seg + 5], (double)(it_tb - it_ta));
        }
    };

    // Two wavefronts per SIMD (KF, IMCC-KF): left alone, issue arbitration serves the older wavefront first -- it finishes its 299 steps in
    // 1.7 ms, the younger one in 2.4-2.8 ms, and runs the last third of its trial without a partner to hide its latencies behind.  The two
    // take turns at the higher priority instead, by the shader clock (a turn = 2^18 cycles = 125 us), told apart by the parity of their
    // wave slot; with equal parities both follow the same schedule and nothing changes.  Measured: residency 0.77 -> 0.86, KF 2.49 -> 2.43 ms.
    unsigned fair_slot = 0;
    constexpr bool FAIR = SHARED_P && L == 2;
    if constexpr (FAIR) {
        unsigned hw_id_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id_));
        fair_slot = hw_id_ & 1u;                                 // WAVE_ID bit 0
    }
    auto store_records = [&](int k) {                            // XREC: the records of step k, straight out of the LDS-resident X
        if constexpr (XREC) {
            // The wavefront's 32 records of this step are 12 KB of contiguous memory.  Store a = 0..3, b = 0..2: lane (tg, pp) = (lane / 8, lane % 8)
            // writes pair 8 b + pp (two consecutive components, 16 bytes) of trial 8 a + tg -- eight trials x 128 contiguous bytes per instruction.
            // The pair sits in one row of X (N is even): LDS cell [r N + j][2 trial + s] and the one 64 doubles further.
            typedef double v2d __attribute__((ext_vector_type(2)));
            if constexpr (XPAIR) {
                // Trial-fastest rows ([step][component][trial], the package's default layout; round 6): the same idea for the OTHER layout.  A row of X in
                // LDS is [lane] = [2 trial + s], so four consecutive doubles are (trial 2u row 2r, trial 2u row 2r + 1, trial 2u + 1 row 2r, trial 2u + 1
                // row 2r + 1) of one column j: two 16-byte pairs of consecutive trials, for the components (2r, j) and (2r + 1, j).  Lane (r, u) = (lane / 16,
                // lane % 16) takes local row r and trial pair u (skewed by 2 r so that the four rows read different LDS banks): per step 6 x (32 bytes out of
                // LDS, two 16-byte stores), every store instruction four runs of 256 contiguous bytes -- 12 stores instead of 24 of 8 bytes.
                const int r4 = (int)(lane >> 4), u = (int)((lane + 2u * (lane >> 4)) & 15u);
                // two cursors walk the six columns of rows 2 r and 2 r + 1 (a component is x_out.sc doubles further): twelve precomputed row addresses would
                // live across the whole step loop -- 24 registers this 256-register kernel does not have
                double *pe = A.x_out.p + ((long long)k * A.x_out.sk + wave_first + 2 * u) + (long long)(2 * r4 * N) * A.x_out.sc;
                double *po = pe + (long long)N * A.x_out.sc;
                const bool live = wave_first + 2 * u < A.T;            // (T is even on this path: a pair is inside the batch or outside it)
                const double *cell = &lds_x[r4 * N][4 * u];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const v2d lo = *reinterpret_cast<const v2d *>(cell + 64 * j), hi = *reinterpret_cast<const v2d *>(cell + 64 * j + 2);
                    v2d even, odd;
                    even.x = lo.x; even.y = hi.x;
                    odd.x = lo.y; odd.y = hi.y;
                    if (live) {
                        *reinterpret_cast<v2d *>(pe) = even;
                        *reinterpret_cast<v2d *>(po) = odd;
                    }
                    pe += A.x_out.sc;
                    po += A.x_out.sc;
                    if (j & 1) __builtin_amdgcn_sched_barrier(0);   // two columns (8 doubles) in flight at a time
                }
                return;
            } else {
            unsigned ln = lane;
            // IMCC-KF sits at its 256-register budget: with the lane-dependent cell / pair offsets below kept across the whole step loop its record instantiation
            // spilled 60 B and ran 8 % slower than this; opaque, they are re-formed every step (a dozen integer instructions).  KF has the registers and keeps them
            // (the same trick costs it 2 %): profiles/r06/pair_stores_ab.txt.
            if constexpr (METHOD == UVS_METHOD_IMCCKF) asm volatile("" : "+v"(ln));
            const int tg = (int)(ln >> 3), pp8 = (int)(ln & 7u);
            double *rec = A.x_out.p + (long long)k * A.x_out.sk + wave_first * (long long)(M * N) + tg * (M * N);
            const double *xl = &lds_x[0][0] + 2 * tg;
#pragma unroll
            for (int b = 0; b < (M * N / 2) / 8; ++b) {
                const int cp = 8 * b + pp8;                              // pair within the record
                const int row = cp / (N / 2), jj = 2 * (cp % (N / 2));   // (lane constants: hoisted out of the step loop by the compiler)
                const int cell = ((row >> 1) * N + jj) * 64 + (row & 1);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    v2d v;
                    v.x = xl[cell + 16 * a];
                    v.y = xl[cell + 64 + 16 * a];
                    if (wave_first + 8 * a + tg < A.T) *reinterpret_cast<v2d *>(rec + (long long)(8 * a) * (M * N) + 2 * cp) = v;
                }
            }
            }
        }
    };
    if constexpr (kDiagFpi) fpi_loop0 = diag_cycles();
    for (int k = k_begin; k < k_end; ++k) {
        asm volatile("" ::: "memory");                           // keep the LDS-resident constants out of loop-invariant hoisting
        // (XREC) the records of the PREVIOUS step leave now: their LDS reads and stores have the whole plant phase to drain under -- issued in one
        // burst behind the rows they stalled the wavefront at the full store queue (measured: RMCKF + 7 %)
        if constexpr (XREC) { if (k > k_begin) store_records(k - 1); }
        UVS_STAMP(5);
        if constexpr (FAIR) {
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();
            if ((((unsigned)(now_ >> kFairPrioLog2)) ^ fair_slot) & 1u) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
        }
        // ---- measurement noise: this step's values were requested a whole step ago; request the next step's now.  vmcnt counts in
        // order, so waiting for a load also waits for every older store: fetched at the top of the step that uses them, the loads sit
        // behind the previous step's err / q stores and the wait in front of the row updates inherits their write latency.
        double nz[R];
#pragma unroll
        for (int r = 0; r < R; ++r) nz[r] = nz_next[r];
        if (on_noise && k + 1 < K) {
#pragma unroll
            for (int r = 0; r < R; ++r) nz_next[r] = pn[r * RS * A.noise.sc];
            pn += A.noise.sk;
        }
        // ---- plant: noise-free features of this lane's rows
        double z[R];
        if constexpr (PLANT == UVS_PLANT_LINEAR) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = r * RS + rb;
                double acc = A.plant.lin_f0[row];
#pragma unroll
                for (int j = 0; j < N; ++j) acc = fma(A.plant.lin_jacobian[row * N + j], q[j] - A.plant.lin_q0[j], acc);
                z[r] = acc;
            }
        } else {
            // sines / cosines of this lane's joint angles.  The joints move by dq * dt per step, so the pair of a joint is carried from step
            // to step by the addition theorems (sincos_advance at the end of the step, 17 instructions per joint) and re-seeded from the
            // angle itself every kSinCosResync steps, or at once when some lane's step was too large for the polynomials (or not finite).
            const double *cj = &lds_c[PC::kJoint + 5 * JG * grp];
            // Whether a lane re-seeds, and from which routine, is decided per lane (the wavefront only shares the branch): a trial's
            // results do not depend on its neighbours in the batch.
            const bool need = reseed || (k & (kSinCosResync - 1)) == 0;
            if (__any(need)) {
                double th[JG], s_new[JG], c_new[JG];
                bool big = false;
#pragma unroll
                for (int u = 0; u < JG; ++u) {
                    th[u] = q[u] + cj[5 * u];
                    big |= !(fabs(th[u]) <= kSinCosBoundedMax);
                    sincos_bounded(th[u], s_new[u], c_new[u]);
                }
                if (__builtin_expect(__any(need && big), 0)) {      // beyond the bounded routine's range (or not finite): the library's reduction
#pragma unroll
                    for (int u = 0; u < JG; ++u) {
                        double sl, cl;
                        sincos(th[u], &sl, &cl);
                        s_new[u] = big ? sl : s_new[u];
                        c_new[u] = big ? cl : c_new[u];
                    }
                }
#pragma unroll
                for (int u = 0; u < JG; ++u) {
                    sn[u] = need ? s_new[u] : sn[u];
                    cs[u] = need ? c_new[u] : cs[u];
                }
                reseed = false;
            }
            double T[3][4];                                 // product of this lane's links (the whole chain when !SPLIT)
            if constexpr (AXIS) {
                // UR10-like chain (kPlantDhAxisAligned): the first link of either lane group has alpha = -pi/2 and the last alpha = 0, so
                // cos / sin alpha are 0, -1 resp. 1, 0 at compile time and the products with them are not written down (38 instructions
                // less per step).  The reference multiplies by cos(-pi/2) = 6.1e-17 instead of 0: a 1e-16 relative difference in the pose.
                {
                    const double s = sn[0], c = cs[0], dd = cj[1], aa = cj[2];
                    T[0][0] = c; T[0][1] = 0.0; T[0][2] = -s; T[0][3] = aa * c;
                    T[1][0] = s; T[1][1] = 0.0; T[1][2] = c; T[1][3] = aa * s;
                    T[2][0] = 0.0; T[2][1] = -1.0; T[2][2] = 0.0; T[2][3] = dd;
                }
                {                                           // middle link: general alpha, against the known zeros of the first
                    const double s = sn[1], c = cs[1];
                    const double dd = cj[5 + 1], aa = cj[5 + 2], ca = cj[5 + 3], sa = cj[5 + 4];
                    const double l01 = -s * ca, l02 = s * sa, l03 = aa * c;
                    const double l11 = c * ca, l12 = -c * sa, l13 = aa * s;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {           // T[r][1] == 0
                        const double t0 = T[r][0], t2 = T[r][2], t3 = T[r][3];
                        T[r][0] = t0 * c;
                        T[r][1] = fma(t0, l01, t2 * sa);
                        T[r][2] = fma(t0, l02, t2 * ca);
                        T[r][3] = fma(t0, l03, fma(t2, dd, t3));
                    }
                    const double t3 = T[2][3];              // row 2 of the first link is (0, -1, 0, d)
                    T[2][0] = -s;
                    T[2][1] = -l11;
                    T[2][2] = -l12;
                    T[2][3] = t3 - l13;
                }
                {                                           // last link: alpha = 0
                    const double s = sn[2], c = cs[2], dd = cj[10 + 1], aa = cj[10 + 2];
                    const double l03 = aa * c, l13 = aa * s;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double t0 = T[r][0], t1 = T[r][1], t2 = T[r][2], t3 = T[r][3];
                        T[r][0] = fma(t0, c, t1 * s);
                        T[r][1] = fma(t1, c, -(t0 * s));
                        T[r][3] = fma(t0, l03, fma(t1, l13, fma(t2, dd, t3)));
                    }
                }
            } else
#pragma unroll
            for (int u = 0; u < JG; ++u) {
                const double s = sn[u], c = cs[u];
                const double dd = cj[5 * u + 1], aa = cj[5 * u + 2], ca = cj[5 * u + 3], sa = cj[5 * u + 4];
                const double l01 = -s * ca, l02 = s * sa, l03 = aa * c;
                const double l11 = c * ca, l12 = -c * sa, l13 = aa * s;
                if (u == 0) {                           // link = Rz(theta) Tz(d) Rx(alpha) Tx(a) (ur10_simulation.py:204-211)
                    T[0][0] = c; T[0][1] = l01; T[0][2] = l02; T[0][3] = l03;
                    T[1][0] = s; T[1][1] = l11; T[1][2] = l12; T[1][3] = l13;
                    T[2][0] = 0.0; T[2][1] = sa; T[2][2] = ca; T[2][3] = dd;
                } else {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double t0 = T[r][0], t1 = T[r][1], t2 = T[r][2], t3 = T[r][3];
                        T[r][0] = fma(t0, c, t1 * s);
                        T[r][1] = fma(t0, l01, fma(t1, l11, t2 * sa));
                        T[r][2] = fma(t0, l02, fma(t1, l12, t2 * ca));
                        T[r][3] = fma(t0, l03, fma(t1, l13, fma(t2, dd, t3)));
                    }
                }
            }
            const double focal = lds_c[PC::kCam], center = lds_c[PC::kCam + 1];
            if constexpr (L == 1) {
#pragma unroll
                for (int pt = 0; pt < M / 2; ++pt) {
                    const double dx = lds_c[PC::kPoint + 3 * pt] - T[0][3], dy = lds_c[PC::kPoint + 3 * pt + 1] - T[1][3],
                                 dz = lds_c[PC::kPoint + 3 * pt + 2] - T[2][3];
                    const double xc = fma(T[0][0], dx, fma(T[1][0], dy, T[2][0] * dz));           // R^T (w - t)
                    const double yc = fma(T[0][1], dx, fma(T[1][1], dy, T[2][1] * dz));
                    const double iz = fast_rcp(fma(T[0][2], dx, fma(T[1][2], dy, T[2][2] * dz)));
                    z[2 * pt] = fma(focal * xc, iz, center);
                    z[2 * pt + 1] = fma(focal * yc, iz, center);
                }
            } else {
                // even lanes own u rows, odd lanes v rows; each lane needs its camera axis (x or y), the optical axis and the position
                const bool odd = sub & 1;
                double va[3], vz[3], vp[3];
                if constexpr (!SPLIT) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) { va[r] = odd ? in_reg(T[r][1]) : in_reg(T[r][0]); vz[r] = T[r][2]; vp[r] = T[r][3]; }
                } else {
                    // right-to-left: start from the last group's columns, then apply the earlier groups' affine maps
                    {
                        double last[3][4];
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 4; ++c) last[r][c] = pair_from<L, G - 1>(T[r][c]);
#pragma unroll
                        for (int r = 0; r < 3; ++r) { va[r] = odd ? in_reg(last[r][1]) : in_reg(last[r][0]); vz[r] = last[r][2]; vp[r] = last[r][3]; }
                    }
#pragma unroll
                    for (int g = G - 2; g >= 0; --g) {
                        double B[3][4];
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 4; ++c) B[r][c] = pair_from_dyn<L>(T[r][c], g);
                        double na[3], nz3[3], np[3];
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            na[r] = fma(B[r][0], va[0], fma(B[r][1], va[1], B[r][2] * va[2]));
                            nz3[r] = fma(B[r][0], vz[0], fma(B[r][1], vz[1], B[r][2] * vz[2]));
                            np[r] = fma(B[r][0], vp[0], fma(B[r][1], vp[1], fma(B[r][2], vp[2], B[r][3])));
                        }
#pragma unroll
                        for (int r = 0; r < 3; ++r) { va[r] = na[r]; vz[r] = nz3[r]; vp[r] = np[r]; }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {                // row r*L + sub looks at point (r*L + sub) / 2
                    double w[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if constexpr (L == 2) w[c] = lds_c[PC::kPoint + 3 * r + c];
                        else if constexpr (EMU2) w[c] = (sub & 2) ? lds_c[PC::kPoint + 3 * (2 + r) + c] : lds_c[PC::kPoint + 3 * r + c];   // row 4 h + 2 r + p: point 2 h + r
                        else w[c] = (sub & 2) ? lds_c[PC::kPoint + 3 * (2 * r + 1) + c] : lds_c[PC::kPoint + 3 * (2 * r) + c];
                    }
                    const double dx = w[0] - vp[0], dy = w[1] - vp[1], dz = w[2] - vp[2];
                    const double ic = fma(va[0], dx, fma(va[1], dy, va[2] * dz));
                    const double iz = fast_rcp(fma(vz[0], dx, fma(vz[1], dy, vz[2] * dz)));
                    z[r] = fma(focal * ic, iz, center);
                }
            }
        }
        UVS_STAMP(0);                                            // noise-load issue + plant
        if constexpr (kDiagStamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); UVS_STAMP(6); }   // slot 6: how long the oldest outstanding memory operation still takes
        const double sigma = bandwidth(fp, k);
        const double neg_half_inv_s2 = -0.5 * fast_rcp(sigma * sigma);
        double c_shared = 1.0;
        if constexpr (METHOD == UVS_METHOD_IMCCKF) {             // one weight for the whole filter: G(||Z - H X||) (experiment.py:258-261)
            double ss = 0.0, nu_im[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double pred = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) pred = fma(XREG ? xr[XREG ? r : 0][j] : lds_x[XREG ? 0 : r * N + j][lane], dq[j], pred);
                const double nu = (z[r] + nz[r] - f_prev[r]) - pred;
                nu_im[r] = nu;
                ss = fma(nu, nu, ss);
            }
            if constexpr (EMU2) {                                // the two-lane chain over four local rows: h = 0 starts it, h = 1 finishes it
                double s1 = dpp_quad<kQuadFromLow>(ss);
#pragma unroll
                for (int r = 0; r < R; ++r) s1 = fma(nu_im[r], nu_im[r], s1);
                ss = emu2_finish(s1);
            } else {
                ss = pair_sum<L>(ss);
            }
            c_shared = exp_nonpos(ss * neg_half_inv_s2);         // sqrt(.)**2 of the reference folded: G(n) = exp(-n^2 / (2 sigma^2))
        }
        double kap[R];
        double chk = 0.0;                                        // turns NaN as soon as any state entry is non-finite
        FpiProbe fpi;
        double pre_nu[R], pre_arg[R];                            // MCKF: innovation and weight argument of the lane's rows, from the pre-pass
        double e_s2[R], e_gg[R], den_emu2 = 0.0;                 // MCKF on four lanes with the two-lane bits: terms of the convergence test
        if constexpr (METHOD == UVS_METHOD_MCKF) {
            mckf_underflow_prepass<R>(fpi, [&](int r) {
                double pred = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) pred = fma(XREG ? xr[XREG ? r : 0][j] : lds_x[XREG ? 0 : r * N + j][lane], dq[j], pred);
                const double nu = ((z[r] + nz[r]) - f_prev[r]) - pred;
                pre_nu[r] = nu;                                   // the rows below take both from here (same operations, same bits: 9 instructions per row less)
                pre_arg[r] = (nu * nu) * neg_half_inv_s2;
                return pre_arg[r];
            });
            if constexpr (EMU2) {                                // ||X||^2 of the convergence test in the two-lane kernel's order (its rows add x.x as they go)
                double d0 = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < N; ++j) d0 = fma(xr[XREG ? r : 0][j], xr[XREG ? r : 0][j], d0);
                double d1 = dpp_quad<kQuadFromLow>(d0);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < N; ++j) d1 = fma(xr[XREG ? r : 0][j], xr[XREG ? r : 0][j], d1);
                den_emu2 = emu2_finish(d1);
            }
            fpi.skip = pair_sum<L>(fpi.skip ? 1.0 : 0.0) != 0.0;  // one underflowed weight anywhere in the filter skips every row's correction
            fpi.skip |= fp.fpi_epoch_max <= 1;                    // "reached max epoch" after the only pass: correction skipped (:246-250)
            fpi.poison = pair_sum<L>(fpi.poison ? 1.0 : 0.0) != 0.0;
            flagged |= alive && (pair_sum<L>(fpi.unsure ? 1.0 : 0.0) != 0.0);
        }
        double m_gamma[R], m_a[R], m_nu[R], m_z[R];              // MCKF: what the undo of a row needs (dead code for the other estimators)
        RowShare<N> share;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double fi = z[r] + nz[r];                      // noisy feature (experiment.py:134-135)
            const double zi = fi - f_prev[r];                    // measurement Z (experiment.py:170-177)
            f_prev[r] = fi;
            double x[N], pb[NP];
#pragma unroll
            for (int j = 0; j < N; ++j) x[j] = XREG ? xr[XREG ? r : 0][j] : lds_x[XREG ? 0 : r * N + j][lane];
            if constexpr (SHARED_P) {                                // one covariance block per lane (KF, IMCC-KF): p[0]
                if (r == 0) {
                    NoHook none;
                    rmckf_row<N, METHOD>(x, p[0], dq, zi, neg_half_inv_s2, c_shared, fp.reg, kap[r], chk, fpi, none, share);
                } else {
                    rmckf_row_follow<N>(x, share, dq, zi, chk);
                    kap[r] = 1.0;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NP; ++e) pb[e] = (r < PV) ? p[r < PV ? r : 0][e] : lds_p[(r >= PV ? r - PV : 0) * NP + e][lane];
                if constexpr (METHOD == UVS_METHOD_MCKF) { fpi.known = true; fpi.known_nu = pre_nu[r]; fpi.known_arg = pre_arg[r]; }
                rmckf_row<N, METHOD>(x, pb, dq, zi, neg_half_inv_s2, c_shared, fp.reg, kap[r], chk, fpi);
            }
            if constexpr (METHOD == UVS_METHOD_MCKF) { m_gamma[r] = fpi.row_gamma; m_a[r] = fpi.row_a; m_nu[r] = fpi.row_nu; m_z[r] = zi; e_s2[r] = fpi.row_s2; e_gg[r] = fpi.row_gg; }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if constexpr (XREG) xr[r][j] = x[j];
                else lds_x[r * N + j][lane] = x[j];
            }
            if constexpr (XOUT && !XREC) {
                {
#pragma unroll
                    for (int j = 0; j < N; ++j) px[(r * RS * N + j) * A.x_out.sc] = x[j];
                }
            }
            if constexpr (!SHARED_P) {
#pragma unroll
                for (int e = 0; e < NP; ++e) {
                    if (r < PV) p[r < PV ? r : 0][e] = pb[e];
                    else lds_p[(r >= PV ? r - PV : 0) * NP + e][lane] = pb[e];
                }
            }
        }
        if constexpr (XOUT && !XREC) px += A.x_out.sk;
        // LDS is the only copy of X from here on: forbid forwarding the stored values into the panel through registers
        asm volatile("" ::: "memory");
        UVS_STAMP(1);                                            // row updates (includes the wait for the noise load)
        if constexpr (METHOD == UVS_METHOD_MCKF && (!XREG || EMU2)) {
            // ---- did the first fixed-point pass settle it (experiment.py:244)?  Rarely not (0.5 % of the steps under Cauchy noise, none
            // for alpha >= 1.3 on the reference's configuration): those lanes take their rows back to the prior state, iterate like
            // Rows::update_mckf and commit the final gain.  The whole wavefront walks through this branch when one of its trials needs it.
            if constexpr (EMU2) {
                // the two-lane kernel's chains over its four local rows: the h = 0 lanes start them, the h = 1 lanes finish them (den_emu2 was
                // formed from the prior X in the pre-pass; the rows left their (gamma nu)^2 and |g|^2 in e_s2 / e_gg)
                double n1 = dpp_quad<kQuadFromLow>(fma(e_s2[1], e_gg[1], fma(e_s2[0], e_gg[0], 0.0)));
                n1 = fma(e_s2[1], e_gg[1], fma(e_s2[0], e_gg[0], n1));
                fpi.num = emu2_finish(n1);
                fpi.den = den_emu2;
            } else {
                fpi.num = pair_sum<L>(fpi.num);
                fpi.den = pair_sum<L>(fpi.den);
            }
            const double thr2 = fp.fpi_threshold * fp.fpi_threshold;
            bool more = alive && !fpi.skip && !fpi.poison && (fpi.num > thr2 * fpi.den);  // ||Xc - X|| / ||X|| > threshold; NaN ends the iteration like the reference's while
            if (__builtin_expect(__any(more), 0)) {
                // ---- the fixed-point branch, spread over the wavefront (round 5).  Round 4 ran it where the state lives: the two lanes of an
                // iterating filter worked through their four rows each while the other 62 lanes of the wavefront waited (a firing cost ~3 800 issue
                // slots; with Cauchy noise one wavefront-step in eight fires, 1.2 filters at a time, one extra pass in 98 % of them).  Here every ROW of
                // an iterating filter gets a lane of its own: a slot of 8 consecutive lanes per filter (up to 8 filters per round, more rounds if more
                // iterate), lane g of a slot taking global row g.  Row state is fetched where it lives -- X and the parked covariance blocks straight
                // from the owner's LDS column, the register-resident blocks, the command and the four undo scalars by ds_bpermute -- the row is undone,
                // factored ONCE (round 4 refactored it every pass), iterated and committed on that lane, and written back the same way.  The only sums
                // over rows, ||Xn - Xc||^2 and ||Xc||^2, are formed in the owner lanes' order: a chain over local rows 0..R-1 handed from lane g to
                // lane g + L of the slot (DPP row_shr), then the sum over the L parities -- bit for bit what round 4 computed (tests/test_gpu_digest.py).
                // EMU2 (four owner lanes per filter, everything in registers): the same branch, with every piece of row state fetched by
                // ds_bpermute; its rows carry the two-lane kernel's numbering (quad lane p + 2 h holds global rows 4 h + 2 r + p), so the chain below
                // -- local rows of a PARITY in order, lane g to lane g + 2 -- is the two-lane kernel's in both mappings.
                static_assert((L == 2 || EMU2) && M <= 8, "slots of 8 lanes; rows interleaved over 2 owner lanes, or the same numbering on a quad");
                const int g = (int)(lane & 7u), slot = (int)(lane >> 3);
                const int rr = EMU2 ? ((g >> 1) & 1) : (g >> 1);                  // owner's local row of global row g ...
                const int own_off = EMU2 ? (g & 1) + 2 * (g >> 2) : (g & 1);      // ... and the owner's position in its lane group
                unsigned long long todo = __ballot(more) & (EMU2 ? 0x1111111111111111ull : 0x5555555555555555ull);   // one bit per iterating filter: its first lane
                UVS_FPI_STAMP(-1);
                if constexpr (kDiagFpi) diag_fpi.sum[6] += 1;
                while (todo) {                                                    // rounds of up to 8 filters (uniform)
                    int src_even = -1, my_slot = -1;
#pragma unroll
                    for (int sidx = 0; sidx < 8; ++sidx) {
                        if (todo) {                                               // uniform
                            const int b = __builtin_ctzll(todo);
                            todo &= todo - 1;
                            src_even = (slot == sidx) ? b : src_even;
                            my_slot = ((int)(lane & ~(unsigned)(L - 1)) == b) ? sidx : my_slot;
                        }
                    }
                    const bool act = src_even >= 0 && g < M;                      // this lane works on a row in this round
                    const int S = (src_even >= 0 ? src_even : (int)(lane & ~7u)) + own_off;   // owner lane of the row (idle slots look at lanes of their own: harmless)
                    const int sa = S << 2;
                    auto pull = [&](int addr, double v) {
                        return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
                    };
                    double *xs = &lds_x[0][0] + ((!XREG && rr < R) ? rr : 0) * (N * 64) + (XREG ? 0 : S);   // the row's X in its owner's LDS column
                    double *ps = &lds_p[0][0] + ((PL > 0 && rr >= PV && rr < R) ? rr - PV : 0) * (NP * 64) + (PL > 0 ? S : 0);   // ... and its parked covariance block (rows >= PV)
                    // Transport in BATCHES: a lone wavefront at the edge of its registers consumes every ds_bpermute result at once (the first version
                    // ran with one or two in flight: 10 k cycles for the 128 pulls of a round, -DUVS_FPI_STAMPS).  The owners' register-resident blocks
                    // are parked in AGPRs FIRST -- that frees the VGPRs the pulls need to overlap -- and are read back one batch at a time as
                    // bpermute sources; each batch issues all its pulls, then selects.
                    ParkedDouble parked[PV > 0 ? PV : 1][NP];
#pragma unroll
                    for (int r = 0; r < PV; ++r)
#pragma unroll
                        for (int e = 0; e < NP; ++e) parked[r][e] = agpr_park(p[r][e]);
                    __builtin_amdgcn_sched_barrier(0);
                    double x[N], pp[NP], h[N], kk[N];
                    // (then the four undo scalars -- their 4 R sources die with the pulls -- and the command)
                    double r_gamma, r_a, r_nu, r_z;
                    {                                                         // the four undo scalars of the row: R candidates each, two batches
                        auto pull2 = [&](const double (&u)[R], const double (&v)[R], double &ru, double &rv) {
                            double cu[R], cv[R];
#pragma unroll
                            for (int r = 0; r < R; ++r) { cu[r] = pull(sa, u[r]); cv[r] = pull(sa, v[r]); }
                            __builtin_amdgcn_sched_barrier(0);
                            ru = cu[0]; rv = cv[0];
#pragma unroll
                            for (int r = 1; r < R; ++r) { ru = (rr == r) ? cu[r] : ru; rv = (rr == r) ? cv[r] : rv; }
                            ru = in_reg(ru); rv = in_reg(rv);                      // (pinned here: LLVM would sink the selects into the row's branch and keep all candidates alive)
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        pull2(m_gamma, m_a, r_gamma, r_a);
                        pull2(m_nu, m_z, r_nu, r_z);
                    }
                    {                                                         // the command (6 doubles)
                        double c[N];
#pragma unroll
                        for (int j = 0; j < N; ++j) c[j] = pull(sa, dq[j]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < N; ++j) h[j] = in_reg(c[j]);
                    }
                    constexpr int PB = 7;                                         // doubles per batch: 14 pulls in flight (the counter holds 15)
#pragma unroll
                    for (int e = 0; e < NP; ++e) pp[e] = (PL > 0) ? ps[e * 64] : 0.0;
#pragma unroll
                    for (int r = 0; r < PV; ++r) {
#pragma unroll
                        for (int e0 = 0; e0 < NP; e0 += PB) {
                            double c[PB];
#pragma unroll
                            for (int i = 0; i < PB; ++i)
                                if (e0 + i < NP) c[i] = pull(sa, agpr_unpark(parked[r][e0 + i]));
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < PB; ++i)
                                if (e0 + i < NP) pp[e0 + i] = in_reg((rr == r) ? c[i] : pp[e0 + i]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (XREG) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            double c[N];
#pragma unroll
                            for (int j = 0; j < N; ++j) c[j] = pull(sa, xr[XREG ? r : 0][j]);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int j = 0; j < N; ++j) x[j] = (r == 0 || rr == r) ? c[j] : x[j];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < N; ++j) x[j] = xs[j * 64];
                    }
                    UVS_FPI_STAMP(0);                                         // slot assignment + parking the owners' blocks + pull-in
                    double Lc[NP], ljj[N];
                    if (act) {
                        mckf_undo_row<N>(x, pp, h, r_gamma, r_a, r_nu, kk);       // back to the prior row and the predicted block
                        mckf_factor_row<N>(pp, Lc, ljj);
                    }
                    UVS_FPI_STAMP(1);                                         // parking the owners' blocks + undo + factor
                    // (the predicted block is not needed while the row iterates: it waits in AGPRs, like the owners' blocks)
                    ParkedDouble pp_parked[NP];
#pragma unroll
                    for (int e = 0; e < NP; ++e) pp_parked[e] = agpr_park(pp[e]);
                    bool skip2 = false, more_t = act;
                    int it_t = 1;
                    while (__any(more_t)) {
                        double kn[N], dd[N], xcv[N];
                        bool bad2 = false;
#pragma unroll
                        for (int l = 0; l < N; ++l) { dd[l] = 0.0; xcv[l] = 0.0; kn[l] = 0.0; }
                        if (more_t) mckf_iterate_row<N>(x, Lc, ljj, h, r_z, neg_half_inv_s2, kk, kn, dd, xcv, bad2);
                        // ||Xn - Xc||^2 and ||Xc||^2 in the owner lanes' order: local rows 0 .. R-1 of a parity as one chain, passed down the slot
                        double cn = 0.0, cd = 0.0;
                        constexpr int CH = M / 2;                                 // local rows of a parity in the two-lane numbering
#pragma unroll
                        for (int st = 0; st < CH; ++st) {
                            double an = cn, ad = cd;
#pragma unroll
                            for (int l = 0; l < N; ++l) { an = fma(dd[l], dd[l], an); ad = fma(xcv[l], xcv[l], ad); }
                            if (st + 1 < CH) { cn = dpp_row_shr<2>(an); cd = dpp_row_shr<2>(ad); }
                            else { cn = an; cd = ad; }
                        }
                        const double num2 = pair_sum<2>(cn), den2 = pair_sum<2>(cd);    // complete on the slot's lanes M - 2, M - 1
                        const unsigned long long zero_rows = __ballot(bad2 && more_t);
                        const bool hit_zero = ((zero_rows >> (lane & ~7u)) & 0xffull) != 0;
                        bool again = false;
                        if (more_t) {
                            if (hit_zero) {                                       // inv(Cy) raises: the correction of this step is skipped (:231-236)
                                skip2 = true;
                            } else {
#pragma unroll
                                for (int j = 0; j < N; ++j) kk[j] = kn[j];
                                ++it_t;
                                if (it_t == fp.fpi_epoch_max) skip2 = true;       // :246-250
                                again = !skip2 && (num2 > thr2 * den2) && it_t < fp.fpi_epoch_max;
                            }
                        }
                        // the verdict of the lanes that hold the complete sums, for the whole slot
                        const unsigned long long verdict = __ballot(again);
                        more_t = more_t && ((verdict >> ((lane & ~7u) | (unsigned)(M - 1))) & 1ull);
                    }
                    UVS_FPI_STAMP(2);                                         // parking the predicted block + the passes
#pragma unroll
                    for (int e = 0; e < NP; ++e) pp[e] = agpr_unpark(pp_parked[e]);
                    if (act) {
                        double unused_chk = 0.0;
                        if (!skip2) mckf_commit_row<N>(x, pp, h, r_z, kk, unused_chk);
                        if constexpr (!XREG) {
#pragma unroll
                            for (int j = 0; j < N; ++j) xs[j * 64] = x[j];
                        }
                        if (PL > 0 && rr >= PV) {
#pragma unroll
                            for (int e = 0; e < NP; ++e) ps[e * 64] = pp[e];
                        }
                        if constexpr (XOUT) {                                     // this step's rows of the X stream: overwrite the optimistic values
                            long long tf = wave_first + src_even / L;
                            tf = tf < A.T ? tf : A.T - 1;
                            double *pxs = A.x_out.at(tf, k, (long long)g * N);
#pragma unroll
                            for (int j = 0; j < N; ++j) pxs[j * A.x_out.sc] = x[j];
                        }
                    }
                    UVS_FPI_STAMP(3);                                         // commit + write-back to LDS + X stream
                    // the register-resident blocks return to their owners
                    const bool is_owner = my_slot >= 0;
                    if constexpr (XREG) {                                         // ... and so does a register-resident X
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const int ta = ((my_slot < 0 ? 0 : my_slot) * 8 + 4 * (sub >> 1) + 2 * r + (sub & 1)) << 2;
#pragma unroll
                            for (int j = 0; j < N; ++j) { const double c = pull(ta, x[j]); xr[XREG ? r : 0][j] = is_owner ? c : xr[XREG ? r : 0][j]; }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < PV; ++r) {
                        const int ta = ((my_slot < 0 ? 0 : my_slot) * 8 + (EMU2 ? 4 * (sub >> 1) + 2 * r + (sub & 1) : r * L + sub)) << 2;
#pragma unroll
                        for (int e0 = 0; e0 < NP; e0 += PB) {                     // batches again: all pulls of a batch in flight, then the selects
                            double c[PB];
#pragma unroll
                            for (int i = 0; i < PB; ++i)
                                if (e0 + i < NP) c[i] = pull(ta, pp[e0 + i]);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < PB; ++i)
                                if (e0 + i < NP) { const double keep = agpr_unpark(parked[r][e0 + i]); p[r][e0 + i] = is_owner ? c[i] : keep; }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                UVS_FPI_STAMP(4);                                             // pull-back of the register-resident blocks (last round)
                asm volatile("" ::: "memory");
            }
        }
        if constexpr (METHOD == UVS_METHOD_MCKF && XREG && !EMU2) {   // plain register-resident variants (L = 1, 4): first pass only, the rest to the careful pass
            fpi.num = pair_sum<L>(fpi.num);
            fpi.den = pair_sum<L>(fpi.den);
            flagged |= alive && fpi_needs_more(fpi, fp);
        }

        // ---- control law: dq = -gain * pinv(X) (kappa o err) (experiment.py:300-312)
        double err[R];
#pragma unroll
        for (int r = 0; r < R; ++r) err[r] = f_prev[r] - des[r]; // experiment.py:302
        double acc_now[3 * R];                                   // statistics accumulators: fetch early, their latency hides under the QR
#pragma unroll
        for (int i = 0; i < 3 * R; ++i) acc_now[i] = lds_acc[i][lane];
        {
            double panel[R][N + 1];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < N; ++j) panel[r][j] = XREG ? xr[XREG ? r : 0][j] : lds_x[XREG ? 0 : r * N + j][lane];
                panel[r][N] = kap[r] * err[r];
            }
            double sol[N];
            // X non-finite: pinv would raise (experiment.py:313-316).  The verdict comes out of the QR's column norms (lstsq_tall_tuned);
            // the per-entry probe `chk` that the rows accumulate is dead code in this kernel.
            bool nonfinite, suspect;
            if constexpr (EMU2) suspect = lstsq_tall_emu2<M, N>(panel, sub, sol, nonfinite);
            else suspect = lstsq_tall_tuned<M, N, L>(panel, sub, sol, nonfinite, METHOD == UVS_METHOD_GMCKF ? (A.fp.reserved & UVS_OPT_STRICT_PINV) != 0 : CERT);
            if constexpr (METHOD == UVS_METHOD_MCKF) nonfinite |= fpi.poison && !fpi.skip;      // the reference's NaN state after a subnormal weight
            if (alive && nonfinite) {
                alive = false;
                status = UVS_STATUS_FAIL;
                k_done = k;
            }
            flagged |= alive && suspect;
#pragma unroll
            for (int j = 0; j < N; ++j) dq[j] = -fp.gain * sol[j];
        }
        if (!__any(alive)) break;                                // the last live trial of the wavefront just FAILed: nothing left to log
        UVS_STAMP(2);                                            // control law

        // ---- logs and statistics
        if (on_err) {
#pragma unroll
            for (int r = 0; r < R; ++r) pe[r * RS * A.err_out.sc] = err[r];
            pe += A.err_out.sk;
        }
        if (on_f) {
#pragma unroll
            for (int r = 0; r < R; ++r) pf[r * RS * A.f_out.sc] = f_prev[r];
            pf += A.f_out.sk;
        }
        double dq_own[JG];                                       // the command for this lane's joints
#pragma unroll
        for (int u = 0; u < JG; ++u) {
            if constexpr (G == 1) dq_own[u] = dq[u];
            else if constexpr (G == 2) dq_own[u] = grp ? in_reg(dq[JG + u]) : in_reg(dq[u]);
            else dq_own[u] = (grp == 0) ? in_reg(dq[u]) : (grp == 1 ? in_reg(dq[JG + u]) : in_reg(dq[2 * JG + u]));
        }
        if (on_q) {
#pragma unroll
            for (int u = 0; u < JG; ++u) pq[u * A.q_out.sc] = q[u];
            pq += A.q_out.sk;
        }
        if (on_dq) {
#pragma unroll
            for (int u = 0; u < JG; ++u) pd[u * A.dq_out.sc] = dq_own[u];
            pd += A.dq_out.sk;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double e = alive ? err[r] : 0.0;               // a failed trial stops contributing (its stats are discarded anyway)
            const double ae = fabs(e);
            lds_acc[r][lane] = fma(e, e, acc_now[r]);
            lds_acc[R + r][lane] = acc_now[R + r] + ae;
            lds_acc[2 * R + r][lane] = fma(t, ae, acc_now[2 * R + r]);
        }
        UVS_STAMP(3);                                            // logs + statistics
        if constexpr (DH) {
            if constexpr (kSinCosStepMax2 > kSinCosStepMax) {
                // Two tiers, chosen PER LANE (the wavefront only shares the branch, so a trial's bits do not depend on its neighbours): steps up
                // to 0.1 rad by the short polynomials, steps up to 1 rad by the long ones, anything else re-seeds from the angle at the next step.
                double dstep[JG], s_old[JG], c_old[JG];
                bool wide = false;
#pragma unroll
                for (int u = 0; u < JG; ++u) {
                    dstep[u] = dq_own[u] * fp.dt;
                    wide |= !(fabs(dstep[u]) <= kSinCosStepMax);
                    reseed |= !(fabs(dstep[u]) <= kSinCosStepMax2);   // too large for either tier, or not finite
                    s_old[u] = sn[u];
                    c_old[u] = cs[u];
                    sincos_advance(sn[u], cs[u], dstep[u]);
                }
                if (__any(wide)) {
#pragma unroll
                    for (int u = 0; u < JG; ++u) {
                        const bool mid = !(fabs(dstep[u]) <= kSinCosStepMax);
                        sincos_advance_wide(s_old[u], c_old[u], dstep[u]);
                        sn[u] = mid ? s_old[u] : sn[u];
                        cs[u] = mid ? c_old[u] : cs[u];
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < JG; ++u) {
                    const double d = dq_own[u] * fp.dt;
                    reseed |= !(fabs(d) <= kSinCosStepMax);             // too large for the polynomials, or not finite: re-seed at the next step
                    sincos_advance(sn[u], cs[u], d);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < JG; ++u) q[u] = fma(dq_own[u], fp.dt, q[u]);       // new_q = q + dq t_s (experiment.py:320)
        t += fp.dt;
    }
    if constexpr (kDiagItems) it_t2 = diag_ticks();
    if constexpr (XREC) { if (k_end > k_begin) store_records(k_end - 1); }   // the last step's records (after a wavefront-wide FAIL: rows past every k_done, unspecified)
    if constexpr (kDiagStamps) {
        diag_steps.sum[4] = diag_ticks() - rt_first;
        if (lane == 0 && A.stats) {
            for (int c = 0; c < 8; ++c) A.stats[3 * wave_first + c] = (double)diag_steps.sum[c];
        }
        return;
    }

    if constexpr (SEG) {
        if (!last_seg) {                                         // hand the chunk to its next segment: state, then the release of the counter
            // An item that fell back (its predecessor's counter did not come within the spin budget and it recomputed the trial from step 0) hands
            // NOTHING over: the chunk's one state slot belongs to the regular chain.  Round 4 let such an item save and publish too; several items
            // of a chunk falling back at once (they are dispatched together, their budgets run out together) then raced for the slot, and a later
            // item could restore a state torn between two of them -- found by tools/fuzz_long.py with the counter withheld at 9 segments (round 5).
            // Its successors run out their own budgets and recompute as well: quadratic work on a path that exists only so that nothing ever hangs.
            const bool fell_back = fresh && seg > 0;
            if (!fell_back) {
            seg_state(std::true_type{});
            __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): every write-through store of the wavefront has been acknowledged before ...
            asm volatile("" ::: "memory");
            // (UVS_OPT_DIAG_DROP_SEG_FLAG: segment 0 keeps its counter to itself, so every later segment runs out its spin budget and recomputes -- tests only)
            if (lane == 0 && !((A.fp.reserved & UVS_OPT_DIAG_DROP_SEG_FLAG) && seg == 0))
                __hip_atomic_store(&A.ws_flags[chunk], seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... the counter moves
            }
            if constexpr (kDiagWaves) record_wave();
            if constexpr (kDiagItems) { it_t3 = diag_ticks(); record_item(true); }
            return;
        }
    }
    if constexpr (kDiagItems) { record_item(false); return; }    // the last segment (or a whole trial): no hand-over
    if constexpr (kDiagFpi) {
        diag_fpi.sum[7] = diag_cycles() - fpi_loop0;             // the whole step loop
        if (lane == 0 && A.stats) {
            for (int c = 0; c < 8; ++c) A.stats[3 * wave_first + c] = (double)diag_fpi.sum[c];
        }
        return;
    }
    double s2[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = lds_acc[c * R + r][lane];
            s2[c] = fma(v, v, s2[c]);
        }
    }
    if constexpr (EMU2) {                                        // the two-lane chain over four local rows, then the pair sum
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s1 = dpp_quad<kQuadFromLow>(s2[c]);
#pragma unroll
            for (int r = 0; r < R; ++r) { const double v = lds_acc[c * R + r][lane]; s1 = fma(v, v, s1); }
            s2[c] = emu2_finish(s1);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) s2[c] = pair_sum<L>(s2[c]);
    }
    if (!valid) return;
    if (sub == 0) {
        if (A.stats && !(kDiagWaves && SEG && A.n_seg > 1)) {   // (wave-times build, segmented: the chunk's slice of `stats` holds the stamps of every segment)
#pragma unroll
            for (int c = 0; c < 3; ++c) A.stats[3 * trial + c] = sqrt(s2[c]);
        }
        if constexpr (kDiagWaves) record_wave();                // overwrites the statistics of the wavefront's first two trials
        if (A.status) A.status[trial] = flagged ? UVS_STATUS_SUSPECT : status;
        if (A.k_done) A.k_done[trial] = k_done;
    }
    if (A.x_final.on()) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j)
                *A.x_final.at(trial, 0, (r * RS + rb) * N + j) = XREG ? xr[XREG ? r : 0][j] : lds_x[XREG ? 0 : r * N + j][lane];
    }
    if (A.p_final.on()) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int l = 0; l < N; ++l)
#pragma unroll
                for (int j = 0; j < N; ++j)
                    *A.p_final.at(trial, 0, ((r * RS + rb) * N + l) * N + j) =
                        SHARED_P ? p[0][Sym<N>::at(l, j)]
                                 : (r < PV) ? p[r < PV ? r : 0][Sym<N>::at(l, j)] : lds_p[(r >= PV ? r - PV : 0) * NP + Sym<N>::at(l, j)][lane];
    }
}

}  // namespace uvs

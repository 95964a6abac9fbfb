// fp64 elementary functions sized for a VALU-bound kernel on gfx950.
//
// The estimator is ~3k fp64 VALU instructions per filter step, so IEEE-exact library division (14 instructions),
// sqrt (~20) and sincos (~70 plus a Payne-Hanek path that drags VGPRs and branches into the loop) are a measurable
// share.  These replacements are accurate to ~1 ulp on the ranges the estimator produces and are validated against
// numpy through uvs_debug_math_f64 (tests/test_gpu_math.py).  They assume normal, finite, in-range inputs; the kernels
// route NaN/Inf through the FAIL path before any result is used.
#pragma once
#include <hip/hip_runtime.h>

namespace uvs {

#ifndef UVS_DEV
#define UVS_DEV __device__ __forceinline__
#endif

// 1/d for normal d: v_rcp_f64 seeds ~2^-23 relative error; two Newton steps reach ~2^-52.
UVS_DEV double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}

// rsqrt(a) and sqrt(a) for normal a > 0 (Goldschmidt-style coupled iteration from v_rsq_f64).
UVS_DEV void fast_sqrt_rsqrt(double a, double &s, double &rs) {
    double y = __builtin_amdgcn_rsq(a);
    double g = a * y;           // ~sqrt(a)
    double h = 0.5 * y;         // ~1/(2 sqrt(a))
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, a);   // final correction of sqrt
    s = fma(d, h, g);
    rs = h + h;
}

// 1/d to ~1e-14 relative (one Newton step): for scale factors whose error only perturbs an orthogonal transform.
UVS_DEV double fast_rcp_1(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    return fma(r, fma(-d, r, 1.0), r);
}

// sqrt(a) to ~1 ulp and 1/sqrt(a) to ~1e-14 relative: one coupled Newton step from v_rsq_f64 plus the final sqrt correction.
UVS_DEV void fast_sqrt_rsqrt_1(double a, double &s, double &rs) {
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    s = fma(fma(-g, g, a), h, g);
    rs = h + h;
}

// exp(x) for x <= 0 (the correntropy weight exp(-e^2 / (2 sigma^2))): Cody-Waite reduction by ln 2, degree-13 Taylor polynomial on
// |r| <= ln2/2 (truncation 4e-18), v_ldexp for the scale (which also produces the denormal / zero tail).  NaN stays NaN; -inf -> 0.
UVS_DEV double exp_nonpos(double x) {
    x = (x < -800.0) ? -800.0 : x;                         // exp underflows to 0 below; keeps the integer conversion in range; NaN passes
    const double kf = rint(x * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                     // 1/13!
    p = fma(p, r, 2.08767569878681e-09);                   // 1/12!
    p = fma(p, r, 2.505210838544172e-08);                  // 1/11!
    p = fma(p, r, 2.755731922398589e-07);                  // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);                 // 1/9!
    p = fma(p, r, 2.48015873015873e-05);                   // 1/8!
    p = fma(p, r, 1.984126984126984e-04);                  // 1/7!
    p = fma(p, r, 1.388888888888889e-03);                  // 1/6!
    p = fma(p, r, 8.333333333333333e-03);                  // 1/5!
    p = fma(p, r, 4.166666666666666e-02);                  // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);                 // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}

// exp(x) for any x with the same kernel: arguments beyond +-800 are clamped (the scale step then gives inf / 0 as exp does).
UVS_DEV double exp_clamped(double x) {
    return exp_nonpos((x > 800.0) ? 800.0 : x);
}

// log(x) for normal positive x (2^-1022 <= x < inf): the classic reduction x = 2^k (1 + f), sqrt(1/2) <= 1 + f < sqrt(2),
// s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-7 minimax R of fdlibm's published coefficients,
// k ln 2 added in two pieces.  33 instructions against the ~85 of the library's extended-precision routine; < 1.5 ulp
// (tests/test_gpu_math.py).  The quotient comes from v_rcp_f64 with one Newton step and one residual correction.
UVS_DEV double log_normal_pos(double x) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                 L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01, L7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);             // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const int up = (m < 7.07106781186547524401e-01) ? 1 : 0;
    m = ldexp(m, up);                                      // [sqrt(1/2), sqrt(2))
    k -= up;
    const double f = m - 1.0;                              // exact
    const double d = 2.0 + f;
    const double r = fast_rcp_1(d);
    double q = f * r;
    q = fma(r, fma(-d, q, f), q);                          // f / (2 + f)
    const double dk = (double)k;
    const double z = q * q, w = z * z;
    const double t1 = w * fma(w, fma(w, L6, L4), L2);
    const double t2 = z * fma(w, fma(w, fma(w, L7, L5), L3), L1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return fma(dk, LN2_HI, -((hfsq - fma(q, hfsq + R, dk * LN2_LO)) - f));
}

// log(x) with numpy's results for every x: the routine above on normal positive arguments, the library's elsewhere (zero, subnormal,
// negative, inf, NaN -- the branch is uniform-false in practice).
UVS_DEV double log_any(double x) {
    const unsigned hi = (unsigned)(__double_as_longlong(x) >> 32);
    if (__builtin_expect(hi - 0x00100000u < 0x7fe00000u, 1)) return log_normal_pos(x);
    return log(x);
}

// sin and cos for |x| <= ~1e5 rad: Cody-Waite reduction by pi/2 in three 33-bit pieces (exact products for
// |k| < 2^20), then the classic minimax kernels on [-pi/4, pi/4] (fdlibm-style coefficients).  < 1 ulp.
UVS_DEV void sincos_bounded(double x, double &s, double &c) {
    const double kf = rint(x * 6.36619772367581382433e-01);
    double r = fma(-kf, 1.57079632673412561417e+00, x);
    r = fma(-kf, 6.07710050630396597660e-11, r);
    // third piece and the rounding error of the second step
    const double w = kf * 2.02226624871116645580e-21;
    const double y = r - w;
    const double yt = (r - y) - w;      // tail
    const double z = y * y;
    // sin kernel
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * y;
    const double ps = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    const double sn = y - ((z * (0.5 * yt - v * ps) - yt) - v * S1);
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double wq = 1.0 - hz;
    const double cs = wq + (((1.0 - wq) - hz) + (z * pc - y * yt));
    const int q = (int)kf & 3;
    const double a = (q & 1) ? cs : sn;
    const double b = (q & 1) ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// Rotation of a tracked (sin, cos) pair by a small angle d, |d| <= kSinCosStepMax: sin(t + d), cos(t + d) from the addition theorems
// with Taylor polynomials in d (truncation < 3e-18 relative at the bound) -- 17 instructions against the 47 of sincos_bounded.  Each
// application adds ~1 ulp of rounding to the pair, so callers re-seed it from the angle itself every kSinCosResync steps.
#ifndef UVS_SINCOS_STEPMAX             // experiment builds: a huge value never re-seeds for a large step (timing only)
#define UVS_SINCOS_STEPMAX 0.1
#endif
constexpr double kSinCosStepMax = UVS_SINCOS_STEPMAX;
constexpr int kSinCosResync = 16;
UVS_DEV void sincos_advance(double &s, double &c, double d) {
    const double z = d * d;
    double ps = 2.7557319223985893e-06;                    // 1/9!
    ps = fma(ps, z, -1.984126984126984e-04);               // -1/7!
    ps = fma(ps, z, 8.333333333333333e-03);                // 1/5!
    ps = fma(ps, z, -1.6666666666666666e-01);              // -1/3!
    const double sd = fma(d * z, ps, d);                   // sin d
    double pc = -2.755731922398589e-07;                    // -1/10!
    pc = fma(pc, z, 2.48015873015873e-05);                 // 1/8!
    pc = fma(pc, z, -1.388888888888889e-03);               // -1/6!
    pc = fma(pc, z, 4.166666666666666e-02);                // 1/4!
    pc = fma(pc, z, -0.5);
    const double m = z * pc;                               // cos d - 1
    const double s1 = s + fma(s, m, c * sd);
    const double c1 = c + fma(c, m, -(s * sd));
    s = s1;
    c = c1;
}

// Second tier: the same rotation for kSinCosStepMax < |d| <= kSinCosStepMax2 (Taylor to d^19 / d^18: truncation < 2e-20 at 1 rad), 27
// instructions.  Config 3 with the outlier hold drives 14 % of the joint steps past 0.1 rad -- 95 % of the wavefront-steps hold such a trial --
// and re-seeds three joints from the angle (47 instructions each) on every one of them.  Built and measured in round 4 (A/B on one box,
// profiles/r04/sincos_tier_ab.txt): config 3 with the hold 12.88 / 12.74 -> 12.68 / 12.63 ms, but config 3 without it 12.03 / 12.22 ->
// 12.17 / 12.39 and the headline 3.149 / 3.145 -> 3.184 / 3.152: keeping the pre-step pair for the per-lane choice costs every step more than
// the held outliers return.  OFF by default (0); -DUVS_SINCOS_STEPMAX2=1.0 builds it.
#ifndef UVS_SINCOS_STEPMAX2
#define UVS_SINCOS_STEPMAX2 0.0
#endif
constexpr double kSinCosStepMax2 = UVS_SINCOS_STEPMAX2;
UVS_DEV void sincos_advance_wide(double &s, double &c, double d) {
    const double z = d * d;
    double ps = -8.2206352466243295e-18;                   // -1/19!
    ps = fma(ps, z, 2.8114572543455206e-15);               // 1/17!
    ps = fma(ps, z, -7.6471637318198164e-13);              // -1/15!
    ps = fma(ps, z, 1.6059043836821613e-10);               // 1/13!
    ps = fma(ps, z, -2.5052108385441720e-08);              // -1/11!
    ps = fma(ps, z, 2.7557319223985893e-06);               // 1/9!
    ps = fma(ps, z, -1.984126984126984e-04);               // -1/7!
    ps = fma(ps, z, 8.333333333333333e-03);                // 1/5!
    ps = fma(ps, z, -1.6666666666666666e-01);              // -1/3!
    const double sd = fma(d * z, ps, d);                   // sin d
    double pc = -1.5619206968586225e-16;                   // -1/18!
    pc = fma(pc, z, 4.7794773323873853e-14);               // 1/16!
    pc = fma(pc, z, -1.1470745597729725e-11);              // -1/14!
    pc = fma(pc, z, 2.08767569878681e-09);                 // 1/12!
    pc = fma(pc, z, -2.755731922398589e-07);               // -1/10!
    pc = fma(pc, z, 2.48015873015873e-05);                 // 1/8!
    pc = fma(pc, z, -1.388888888888889e-03);               // -1/6!
    pc = fma(pc, z, 4.166666666666666e-02);                // 1/4!
    pc = fma(pc, z, -0.5);
    const double m = z * pc;                               // cos d - 1
    const double s1 = s + fma(s, m, c * sd);
    const double c1 = c + fma(c, m, -(s * sd));
    s = s1;
    c = c1;
}

// Range at which sincos_bounded hands over to the library routine (exact huge-argument reduction).
constexpr double kSinCosBoundedMax = 1.0e5;

UVS_DEV void sincos_any(double x, double &s, double &c) {
    if (__builtin_expect(fabs(x) <= kSinCosBoundedMax, 1)) {
        sincos_bounded(x, s, c);
    } else {
        sincos(x, &s, &c);
    }
}

}  // namespace uvs

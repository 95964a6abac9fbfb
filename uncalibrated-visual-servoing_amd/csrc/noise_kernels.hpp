// On-device measurement-noise streams, value-compatible with the reference's NoiseProfiler (noise.py:29-207) and hence with
// numpy.random.Generator(PCG64): SURVEY.md section 8f rank 1.
//
// Restated third-party algorithms (numpy is not vendored in the reference; requirements.txt:1 pins numpy==2.2.4):
//   * PCG64 (numpy/random/src/pcg64/pcg64.h): 128-bit LCG state = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc, output
//     XSL-RR: rotr64(hi ^ lo, hi >> 58) of the *advanced* state; next_double = (u64 >> 11) * 2^-53.  Seeding (SeedSequence +
//     pcg64_set_seed) is pcg64_seed_kernel below (pcg.py holds the same arithmetic in numpy as the host check).
//   * Generator.uniform(low, high) = low + (high - low) * next_double;  Generator.normal(loc, scale) = loc + scale * z with z
//     from the 256-layer ziggurat of numpy/random/src/distributions/distributions.c (random_standard_normal): tables
//     fi / wi / ki shipped in data/ziggurat_normal.npz (tools/extract_ziggurat_tables.py), tail and wedge tests as there.
// One lane owns one feature *pair* of one trial, because the outlier hold couples the two features of a pair
// (noise.py:82-116); lanes of a wavefront hold consecutive trials so the trial-fastest output layout is written coalesced.
#pragma once
#include "rmckf_math.hpp"
#include "rmckf_device.hpp"

namespace uvs {

struct NoiseArgs {
    uvs_noise_params np;
    long long T;
    const unsigned long long *states;      // [T][n_gen][4]: state_hi, state_lo, inc_hi, inc_lo
    const double *zig;                     // fi[256], wi[256], ki[256] (ki as raw u64 bits)
    View out;
    int chunks = 1, draws_per_sample = 0;  // noise_streams_kernel: lanes per stream and PCG64 draws per sample (0 = not fixed: one chunk)
};

struct Pcg64 {
    unsigned long long sh, sl, ih, il;
    UVS_DEV void load(const unsigned long long *p) { sh = p[0]; sl = p[1]; ih = p[2]; il = p[3]; }
    UVS_DEV unsigned long long next64() {
        const unsigned long long MH = 0x2360ED051FC65DA4ULL, ML = 0x4385DF649FCCF645ULL;
        unsigned long long lo = sl * ML;
        unsigned long long hi = __umul64hi(sl, ML) + sl * MH + sh * ML;
        const unsigned long long nlo = lo + il;
        hi += ih + (nlo < lo ? 1ULL : 0ULL);
        sh = hi; sl = nlo;
        const unsigned long long x = hi ^ nlo;
        const unsigned rot = (unsigned)(hi >> 58);
        return (x >> rot) | (x << ((64u - rot) & 63u));
    }
    UVS_DEV double next_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
    // The generator `delta` draws further on, in O(log delta) 128-bit multiply-adds (the LCG jump of pcg64.h, pcg_advance_lcg_128): lets several
    // lanes share ONE stream, each starting at its own step (noise_streams_kernel).
    UVS_DEV void advance(unsigned long long delta) {
        auto mul = [](unsigned long long ah, unsigned long long al, unsigned long long bh, unsigned long long bl, unsigned long long &rh, unsigned long long &rl) {
            rl = al * bl;
            rh = __umul64hi(al, bl) + al * bh + ah * bl;
        };
        unsigned long long cmh = 0x2360ED051FC65DA4ULL, cml = 0x4385DF649FCCF645ULL, cph = ih, cpl = il;      // current multiplier / increment
        unsigned long long amh = 0, aml = 1, aph = 0, apl = 0;                                                // accumulated multiplier / increment
        while (delta) {
            if (delta & 1ULL) {
                mul(amh, aml, cmh, cml, amh, aml);
                unsigned long long th, tl;
                mul(aph, apl, cmh, cml, th, tl);
                apl = tl + cpl;
                aph = th + cph + (apl < tl ? 1ULL : 0ULL);
            }
            unsigned long long mh = cmh, ml = cml + 1ULL;                                                       // (cur_mult + 1) * cur_plus
            mh += (ml == 0ULL) ? 1ULL : 0ULL;
            mul(mh, ml, cph, cpl, cph, cpl);
            mul(cmh, cml, cmh, cml, cmh, cml);
            delta >>= 1;
        }
        unsigned long long th, tl;
        mul(amh, aml, sh, sl, th, tl);
        sl = tl + apl;
        sh = th + aph + (sl < tl ? 1ULL : 0ULL);
    }
};

// numpy's integer seed -> PCG64 stream, on the device.  SeedSequence (numpy/random/bit_generator.pyx): the seed's 32-bit words
// (one, or two when seed >= 2^32) are hashed into a pool of 4 words, every pool word is mixed into every other, and
// generate_state(4, uint64) hashes the pool out again; pcg64_set_seed then runs state = 0; step; state += initstate; step
// with inc = (initseq << 1) | 1.  One generator per thread; a sweep needs half a million of them (noise.py:59,70).
__global__ __launch_bounds__(256) void pcg64_seed_kernel(long long n, const unsigned long long *seeds, unsigned long long *states) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long seed = seeds[i];
    const unsigned e0 = (unsigned)seed, e1 = (unsigned)(seed >> 32);
    unsigned hc = 0x43b0d7e5u;
    auto hashmix = [&hc](unsigned v) {
        v ^= hc;
        hc *= 0x931e8875u;
        v *= hc;
        return v ^ (v >> 16);
    };
    auto mix = [](unsigned x, unsigned y) {
        const unsigned r = 0xca01f9ddu * x - 0x4973f715u * y;
        return r ^ (r >> 16);
    };
    unsigned pool[4];
    pool[0] = hashmix(e0);
    pool[1] = hashmix(e1);                                          // a missing second word hashes as 0, and e1 == 0 then
    pool[2] = hashmix(0u);
    pool[3] = hashmix(0u);
#pragma unroll
    for (int src = 0; src < 4; ++src)
#pragma unroll
        for (int dst = 0; dst < 4; ++dst)
            if (src != dst) pool[dst] = mix(pool[dst], hashmix(pool[src]));
    unsigned hb = 0x8b51f9ddu, w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned d = pool[k & 3] ^ hb;
        hb *= 0x58f38dedu;
        d *= hb;
        w[k] = d ^ (d >> 16);
    }
    const unsigned long long init_hi = w[0] | ((unsigned long long)w[1] << 32), init_lo = w[2] | ((unsigned long long)w[3] << 32);
    const unsigned long long seq_hi = w[4] | ((unsigned long long)w[5] << 32), seq_lo = w[6] | ((unsigned long long)w[7] << 32);
    const unsigned long long inc_hi = (seq_hi << 1) | (seq_lo >> 63), inc_lo = (seq_lo << 1) | 1ULL;
    const unsigned long long MH = 0x2360ED051FC65DA4ULL, ML = 0x4385DF649FCCF645ULL;
    unsigned long long sl = inc_lo + init_lo;                        // (0 * mult + inc) + initstate
    unsigned long long sh = inc_hi + init_hi + (sl < inc_lo ? 1ULL : 0ULL);
    const unsigned long long lo = sl * ML;
    unsigned long long hi = __umul64hi(sl, ML) + sl * MH + sh * ML;
    const unsigned long long nlo = lo + inc_lo;
    hi += inc_hi + (nlo < lo ? 1ULL : 0ULL);
    states[4 * i + 0] = hi;
    states[4 * i + 1] = nlo;
    states[4 * i + 2] = inc_hi;
    states[4 * i + 3] = inc_lo;
}

UVS_DEV double standard_normal(Pcg64 &g, const double *zig) {
    const double *fi = zig, *wi = zig + 256;
    const unsigned long long *ki = reinterpret_cast<const unsigned long long *>(zig + 512);
    const double R = 3.6541528853610087963519472518, INV_R = 0.27366123732975827203338247596;
    for (;;) {
        unsigned long long r = g.next64();
        const int idx = (int)(r & 0xff);
        r >>= 8;
        const int sign = (int)(r & 1);
        const unsigned long long rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * wi[idx];
        if (sign) x = -x;
        if (rabs < ki[idx]) return x;                                   // ~99.3 %: inside the layer
        if (idx == 0) {                                                 // tail
            for (;;) {
                const double xx = -INV_R * log1p(-g.next_double());
                const double yy = -log1p(-g.next_double());
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(R + xx) : R + xx;
            }
        } else if ((fi[idx - 1] - fi[idx]) * g.next_double() + fi[idx] < exp(-0.5 * x * x)) {
            return x;                                                   // wedge
        }
    }
}

// One fresh sample of feature `i` (noise.py:120-207).  gens[]: this feature's main generators (component 0, 1, 2), sel: selector.
// TYPE: the noise type, fixed at compile time (each instantiation carries only the generators and the code of its own type; with the type
// chosen at run time every lane held 8 generator states = 64 registers and the kernel ran two wavefronts per SIMD).
// Kernel-internal noise type: ALPHA_STABLE with beta = 0 and alpha not in {0.5 (|beta| = 1), 1, 2} -- the reference's own sweeps
// (config.json: alpha = linspace(1, 2, 12), beta = 0).  The launcher selects it; the step loop then carries none of the run-time
// case analysis on (alpha, beta), which costs the general instantiation registers, scalar moves and branches on every sample.
constexpr int kNoiseStableSymmetric = 6;
// Kernel-internal noise type: ALPHA_STABLE with the Chambers-Mallows-Stuck transform evaluated AS noise.py:188-199 WRITES IT -- three library
// sin / cos, two library pow(), library log, the reference's order of operations -- instead of the folded exponential of the default kernels.
// Selected by UVS_NOISE_OPT_AS_WRITTEN in uvs_noise_params.type: closer to numpy's bits in the far tails (tools/fuzz_noise.py --ulp), slower.
constexpr int kNoiseStableAsWritten = 7;

// Whether the symmetric instantiation's addition-theorem cosine is accurate enough for these parameters (see draw_stable_symmetric).
inline bool stable_symmetric_fast(const uvs_noise_params &p) {
    if (!(p.beta == 0.0 && p.alpha != 1.0 && p.alpha != 2.0 && p.alpha > 0.0 && p.alpha < 2.0)) return false;   // (alpha = 0.5 is special only with |beta| = 1: noise.py:185)
    const double floor_c2 = cos(fabs(1.0 - p.alpha) * 1.5707963267948966);
    return fabs(p.expo) * 2.0e-16 <= 2.0e-14 * floor_c2;
}

// The general Chambers-Mallows-Stuck draw for beta = 0 (noise.py:188-192): see the comments in draw<> below.
UVS_DEV double draw_stable_symmetric(const uvs_noise_params &p, Pcg64 &g) {
    const double HALF_PI = 1.5707963267948966, PI = 3.141592653589793;
    const double V = -HALF_PI + PI * g.next_double();
    const double W = -log_any(0.0 + 1.0 * g.next_double());
    double s1, c1, s0, c0;
    sincos_bounded(p.alpha * V, s1, c1);
    sincos_bounded(V, s0, c0);
    // cos((1 - alpha) V) = cos(V - alpha V) by the addition theorem: two instructions instead of a third sincos.  Its absolute error of
    // ~2e-16 is a relative one of 2e-16 / cos(|1 - alpha| pi / 2) at worst, which enters the sample times |(1 - alpha) / alpha|; the launcher
    // selects this kernel only where that product stays below 2e-14 (stable_symmetric_fast below), a tenth of the noise fixtures' gate.
    const double c2 = fma(s0, s1, c0 * c1);
    const double ratio = __builtin_expect(W <= 1.0e300, 1) ? c2 * fast_rcp(W) : c2 / W;
    const double e = exp_clamped(fma(p.expo, log_any(ratio), -p.inv_alpha * log_any(c0)));
    return p.gamma * (s1 * e) + p.delta;
}

template <int TYPE>
UVS_DEV double draw(const uvs_noise_params &p, Pcg64 *gens, Pcg64 &sel, const double *zig) {
    const double HALF_PI = 1.5707963267948966, PI = 3.141592653589793;
    if constexpr (TYPE == kNoiseStableSymmetric) return draw_stable_symmetric(p, gens[0]);
    constexpr bool AS_WRITTEN = (TYPE == kNoiseStableAsWritten);
    switch (TYPE) {
        case UVS_NOISE_WHITE: return 0.0 + p.std * standard_normal(gens[0], zig);
        case UVS_NOISE_UNIFORM: return gens[0].next_double();                              // uniform(): 0 + 1 * u
        case UVS_NOISE_GAUSSIAN_MIXTURE: {
            const double u = 0.0 + 1.0 * sel.next_double();
            if (u > p.rho) return 0.0 + p.std * standard_normal(gens[0], zig);
            return p.mean + p.std * standard_normal(gens[1], zig);
        }
        case UVS_NOISE_GAUSSIAN_BIMODAL: {
            const double u = 0.0 + 1.0 * sel.next_double();
            if (u > p.rho) return 0.0 + p.std * standard_normal(gens[0], zig);
            if (u > p.rho / 2) return p.mean + p.std * standard_normal(gens[1], zig);
            return -p.mean + p.std * standard_normal(gens[2], zig);
        }
        default: break;
    }
    // ALPHA_STABLE: Chambers-Mallows-Stuck with the reference's special cases (noise.py:179-205)
    double x;
    if (p.alpha == 2.0) {
        x = 0.0 + p.sqrt2 * standard_normal(gens[0], zig);
    } else if (p.alpha == 1.0 && p.beta == 0.0) {
        x = tan(-HALF_PI + PI * gens[0].next_double());
    } else if (p.alpha == 0.5 && fabs(p.beta) == 1.0) {
        const double z = 0.0 + 1.0 * standard_normal(gens[0], zig);
        x = p.beta / (z * z);
    } else {
        const double V = -HALF_PI + PI * gens[0].next_double();
        const double u_w = 0.0 + 1.0 * gens[0].next_double();
        const double W = AS_WRITTEN ? -log(u_w) : -log_any(u_w);           // 1.1e-16 <= W <= 36.8, or +inf for a zero draw (probability 2^-53)
        if (AS_WRITTEN && p.alpha != 1.0) {
            // noise.py:188-199 operation for operation (1 / alpha, (1 - alpha) / alpha, 1 - alpha come from the host's Python floats)
            if (p.beta == 0.0) x = (sin(p.alpha * V) / pow(cos(V), p.inv_alpha)) * pow(cos(V * p.one_minus_alpha) / W, p.expo);
            else x = p.cms_S * sin(p.alpha * V + p.cms_B) / pow(cos(V), p.inv_alpha) * pow(cos(p.one_minus_alpha * V - p.cms_B) / W, p.expo);
        } else if (p.alpha != 1.0) {
            // sin(aV + B) / cos(V)^(1/a) * (cos((1-a)V - B) / W)^((1-a)/a), B = 0 and S = 1 when beta = 0 (noise.py:188-199).  The two powers
            // are folded into one exponential, exp(e log(c2 / W) - log(c1) / a): three bounded-argument sincos (|angle| < 3 pi / 2), two logs
            // and one exp instead of three trigonometric calls and two pow(); differs from the reference's evaluation by
            // <= (|exponent| + 2) ulp (|exponent| <~ 40 even for the 1e-16 tails of cos V and W), inside the 2e-13 gate of the noise fixtures.
            // The logs and the exp are this library's own short routines (rmckf_math.hpp: 33 and 21 instructions against ~85 and ~30 of the
            // extended-precision library ones; same <= 1.5 ulp); zero / negative / non-finite arguments take the library's.
            // A negative base (possible only with beta != 0) gives NaN through log exactly as numpy's pow does.
            const bool sym = (p.beta == 0.0);
            const double B = sym ? 0.0 : p.cms_B;
            double s1, c1, s0, c0, s2, c2;
            sincos_bounded(sym ? p.alpha * V : p.alpha * V + B, s1, c1);
            sincos_bounded(V, s0, c0);
            sincos_bounded(sym ? V * p.one_minus_alpha : p.one_minus_alpha * V - B, s2, c2);
            // (c2 / W as c2 * (1 / W): W is normal and positive unless the draw was zero, which takes the division)
            const double ratio = __builtin_expect(W <= 1.0e300, 1) ? c2 * fast_rcp(W) : c2 / W;
            const double e = exp_clamped(fma(p.expo, log_any(ratio), -p.inv_alpha * log_any(c0)));
            x = sym ? s1 * e : p.cms_S * s1 * e;
        } else {
            const double sv = HALF_PI + p.beta * V;
            x = p.two_over_pi * (sv * tan(V) - p.beta * log((W * cos(V)) / sv));
        }
    }
    return (p.alpha == 1.0) ? p.gamma * x + p.shift + p.delta : p.gamma * x + p.delta;
}

#ifdef UVS_NOISE_WAVES                  // experiment builds: pin the occupancy the register allocator aims at
#define UVS_NOISE_OCC __attribute__((amdgpu_waves_per_eu(UVS_NOISE_WAVES, UVS_NOISE_WAVES)))
#else
#define UVS_NOISE_OCC
#endif
template <int TYPE>
__global__ __launch_bounds__(64) UVS_NOISE_OCC void noise_kernel(const NoiseArgs A) {
    const uvs_noise_params &p = A.np;
    const int pairs = p.m / 2;
    const long long gid = (long long)blockIdx.x * 64 + threadIdx.x;
    if (gid >= A.T * pairs) return;
    const int pair = (int)(gid / A.T);                       // trial fastest: a wavefront holds one pair of 64 consecutive trials
    const long long t = gid % A.T;
    constexpr int comps = (TYPE == UVS_NOISE_GAUSSIAN_MIXTURE) ? 2 : (TYPE == UVS_NOISE_GAUSSIAN_BIMODAL ? 3 : 1);
    constexpr bool has_sel = comps > 1;
    const int n_gen = comps * p.m + (has_sel ? p.m : 0);
    Pcg64 g[2][comps], sel[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = 2 * pair + h;
#pragma unroll
        for (int c = 0; c < comps; ++c) g[h][c].load(A.states + ((t * n_gen) + i + c * p.m) * 4);                 // generators[i + c*m], noise.py:134-148
        if constexpr (has_sel) sel[h].load(A.states + ((t * n_gen) + comps * p.m + i) * 4);                       // rhoGenerators[i], noise.py:59
        else sel[h].sh = sel[h].sl = sel[h].ih = sel[h].il = 0;                                                   // unused
    }
    double cur0 = 0.0, cur1 = 0.0;
    int cnt = 0, cnt_max = 0;                                 // noise_hold_cnt / noise_hold_cnt_max of this pair
    for (int k = 0; k < p.steps; ++k) {
        if (cnt >= cnt_max) {                                 // noise.py:83-113
            cnt = 0;
            cur0 = draw<TYPE>(p, g[0], sel[0], A.zig);
            cur1 = draw<TYPE>(p, g[1], sel[1], A.zig);
            cnt_max = (fabs(cur0) > 20.0 || fabs(cur1) > 20.0) ? p.hold_cnt : 0;
        } else {
            ++cnt;                                            // noise.py:114-116
        }
        *A.out.at(t, k, 2 * pair) = cur0;
        *A.out.at(t, k, 2 * pair + 1) = cur1;
    }
}

// One generator per lane, no hold: stream s is the getNoise() sequence of a feature whose generator is PCG64 in state A.states[s] -- for the
// noise types with one generator per feature (noise.py:66-70: WHITE_NOISE, ALPHA_STABLE, UNIFORM).  The Monte-Carlo driver seeds feature i of
// trial t with seed0 + t + 10 i (main.py:137-139 + noise.py:70), so feature i + 1 of trial t IS feature i of trial t + 10: a cell of T trials
// holds only T + 10 (m - 1) distinct streams, and the closed loop reads them through a view with trial_stride 1, comp_stride 10 (SURVEY.md
// section 7; uvs_noise_generate_streams_f64).  Rows are stream-fastest: a wavefront writes 512 contiguous bytes per step.
template <int TYPE>
__global__ __launch_bounds__(64) UVS_NOISE_OCC void noise_streams_kernel(const NoiseArgs A) {
    static_assert(TYPE == UVS_NOISE_WHITE || TYPE == UVS_NOISE_ALPHA_STABLE || TYPE == UVS_NOISE_UNIFORM || TYPE == kNoiseStableSymmetric ||
                  TYPE == kNoiseStableAsWritten, "one generator per feature");
    const uvs_noise_params &p = A.np;
    const long long gid = (long long)blockIdx.x * 64 + threadIdx.x;
    // A.chunks lanes share a stream (stream fastest: a wavefront still writes 512 contiguous bytes per step), each generating its own range of
    // steps from the generator jumped ahead -- only where a sample takes a FIXED number of draws (A.draws_per_sample: 2 for the Chambers-Mallows-
    // Stuck transform, 1 for uniform and Cauchy; the ziggurat normal does not, and runs with one chunk).  T + 70 streams are 1 026 wavefronts, one
    // per SIMD and nothing to overlap with: in four chunks the same work runs four wavefronts deep.
    const long long s = gid % A.T;
    const int chunk = (int)(gid / A.T);
    if (chunk >= A.chunks) return;
    const int per = (p.steps + A.chunks - 1) / A.chunks, k0 = chunk * per, k1 = (k0 + per < p.steps) ? k0 + per : p.steps;
    Pcg64 g[1], sel;
    g[0].load(A.states + s * 4);
    sel.sh = sel.sl = sel.ih = sel.il = 0;                    // unused
    if (k0 > 0) g[0].advance((unsigned long long)A.draws_per_sample * (unsigned long long)k0);
    double *o = A.out.p + s * A.out.st + (long long)k0 * A.out.sk;
    for (int k = k0; k < k1; ++k) {
        *o = draw<TYPE>(p, g, sel, A.zig);
        o += A.out.sk;
    }
}

}  // namespace uvs

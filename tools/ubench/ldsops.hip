// Issue cost of LDS instructions for a lone wavefront per SIMD: 12 LDS operations spread over / bunched before 768 FMAs.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/ldsops.hip -o tools/ubench/ldsops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double2v __attribute__((ext_vector_type(2)));
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R12(x) R4(x) R4(x) R4(x)
#define R6(x) x x x x x x
// operands: %0 d2 (4 VGPRs), %1-%3 accumulators, %4 dr (2 VGPRs, read target), %5 b, %6 c, %7 d1 (2 VGPRs), %8 lds address lane*8, %9 lane*16
#define FMA8 "v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n"
#define FMA64 R4(FMA8) R4(FMA8)
#define FMA768 R12(FMA64)

#define KERNEL(NAME, BODY)                                                                                         \
    __global__ __launch_bounds__(64) void NAME(double *out, unsigned long long *cyc, double seed) {                 \
        __shared__ double buf[64 * 64];                                                                            \
        double2v d2 = {seed + threadIdx.x, seed - threadIdx.x};                                                     \
        double dr = seed, d1 = seed * threadIdx.x, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = 1.0000001, c = 1e-9;       \
        for (int i = 0; i < 64; ++i) buf[i * 64 + threadIdx.x] = seed;                                             \
        unsigned l8 = threadIdx.x * 8, l16 = threadIdx.x * 16;                                                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                        \
        for (int it = 0; it < 64; ++it) {                                                                          \
            asm volatile(BODY "s_waitcnt lgkmcnt(0)\n" : "+v"(d2), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(dr) : "v"(b), "v"(c), "v"(d1), "v"(l8), "v"(l16) : "memory"); \
        }                                                                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                        \
        out[(size_t)blockIdx.x * 64 + threadIdx.x] = a1 + a2 + a3 + d2.x + dr + buf[threadIdx.x];                       \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                           \
    }

KERNEL(k_none, FMA768)
KERNEL(k_w64_spread, R12("ds_write_b64 %8, %7 offset:512\n" FMA64))
KERNEL(k_w64_burst, R12("ds_write_b64 %8, %7 offset:512\n") FMA768)
KERNEL(k_w2_spread, R12("ds_write2st64_b64 %8, %7, %7 offset0:2 offset1:3\n" FMA64))
KERNEL(k_w2_burst, R12("ds_write2st64_b64 %8, %7, %7 offset0:2 offset1:3\n") FMA768)
KERNEL(k_w128_spread, R12("ds_write_b128 %9, %0 offset:4096\n" FMA64))
KERNEL(k_w128_burst, R12("ds_write_b128 %9, %0 offset:4096\n") FMA768)
KERNEL(k_r64_spread, R12("ds_read_b64 %4, %8 offset:512\n" FMA64))
KERNEL(k_r2_spread, R12("ds_read2st64_b64 %0, %8 offset0:2 offset1:3\n" FMA64))
KERNEL(k_r2_burst, R12("ds_read2st64_b64 %0, %8 offset0:2 offset1:3\n") FMA768)
KERNEL(k_r128_spread, R12("ds_read_b128 %0, %9 offset:4096\n" FMA64))
KERNEL(k_r128_burst, R12("ds_read_b128 %0, %9 offset:4096\n") FMA768)

typedef void (*kern_t)(double *, unsigned long long *, double);
int main(int argc, char **argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    struct { const char *name; kern_t fn; } tab[] = {
        {"no LDS", k_none}, {"12 ds_write_b64 spread", k_w64_spread}, {"12 ds_write_b64 burst", k_w64_burst},
        {"12 ds_write2st64_b64 spread", k_w2_spread}, {"12 ds_write2st64_b64 burst", k_w2_burst},
        {"12 ds_write_b128 spread", k_w128_spread}, {"12 ds_write_b128 burst", k_w128_burst},
        {"12 ds_read_b64 spread", k_r64_spread}, {"12 ds_read2st64_b64 spread", k_r2_spread}, {"12 ds_read2st64_b64 burst", k_r2_burst},
        {"12 ds_read_b128 spread", k_r128_spread}, {"12 ds_read_b128 burst", k_r128_burst}};
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024;             // 1024 = one wavefront per SIMD; 256 / 512: one / two wavefronts per CU
    printf("%d wavefronts\n", blocks);
    double base = 0;
    for (auto &e : tab) {
        double *out; unsigned long long *cyc;
        (void)hipMalloc(&out, blocks * 64 * 8); (void)hipMalloc(&cyc, blocks * sizeof(unsigned long long));
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.5);
        hipError_t err = hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        const double per = s / blocks / 64.0;
        if (e.fn == k_none) base = per;
        printf("  %-34s %8.1f cycles per group  (+%.1f per LDS instruction) %s\n", e.name, per, (per - base) / 12.0, err == hipSuccess ? "" : hipGetErrorString(err));
        (void)hipFree(out); (void)hipFree(cyc);
    }
    return 0;
}

// How long does one streaming store hold up a lone wavefront?  One store per 127 FMAs (bandwidth far below the HBM limit),
// several encodings.  build: hipcc -O3 --offload-arch=gfx950 tools/ubench/stores.hip -o tools/ubench/stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define FMA8 "v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n"
#define FMA128 R16(FMA8)

// %0 a0 (data), %1-%3 accumulators, %4 b, %5 c, %6 64-bit per-lane pointer (advanced by s[20:21] bytes), %7 32-bit lane offset
#define KERNEL(NAME, STORE)                                                                                        \
    __global__ __launch_bounds__(64) void NAME(double *out, unsigned long long *cyc, double seed, long long stride) { \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 1e-9;            \
        double *gp = out + (size_t)blockIdx.x * 64 + threadIdx.x;                                                  \
        unsigned voff = threadIdx.x * 8;                                                                           \
        unsigned long long sbase = (unsigned long long)(out + (size_t)blockIdx.x * 64);                            \
        asm volatile("s_mov_b32 s20, %0\n s_mov_b32 s21, %1\n s_mov_b32 s22, %2\n s_mov_b32 s23, %3\n"               \
                     "s_mov_b32 s24, %2\n s_mov_b32 s25, %3\n s_mov_b32 s26, 0x7fffffff\n s_mov_b32 s27, 0x00020000\n" \
                     :: "s"((unsigned)stride), "s"((unsigned)(stride >> 32)), "s"((unsigned)sbase), "s"((unsigned)(sbase >> 32)) \
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                        \
        for (int it = 0; it < 256; ++it) {                                                                         \
            asm volatile(STORE FMA128 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(gp) : "v"(b), "v"(c), "v"(voff) \
                         : "memory", "scc", "s22", "s23", "s24", "s25");                                             \
        }                                                                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                        \
        out[(size_t)blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;                                            \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                           \
    }
// operand numbering inside STORE: %0 data, %4 pointer(gp), %5.. shift by the "+v" list: outputs %0-%4, inputs %5 b, %6 c, %7 voff
#undef FMA8
#define FMA8 "v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n"

KERNEL(k_none, "v_lshl_add_u64 %4, %4, 0, s[20:21]\n")
KERNEL(k_x2, "global_store_dwordx2 %4, %0, off\n v_lshl_add_u64 %4, %4, 0, s[20:21]\n")
KERNEL(k_x2_nt, "global_store_dwordx2 %4, %0, off nt\n v_lshl_add_u64 %4, %4, 0, s[20:21]\n")
KERNEL(k_x2_sc1, "global_store_dwordx2 %4, %0, off sc1\n v_lshl_add_u64 %4, %4, 0, s[20:21]\n")
KERNEL(k_x2_sc0sc1, "global_store_dwordx2 %4, %0, off sc0 sc1\n v_lshl_add_u64 %4, %4, 0, s[20:21]\n")
KERNEL(k_x1, "global_store_dword %4, %7, off\n v_lshl_add_u64 %4, %4, 0, s[20:21]\n")
KERNEL(k_saddr, "global_store_dwordx2 %7, %0, s[22:23]\n s_add_u32 s22, s22, s20\n s_addc_u32 s23, s23, s21\n")
KERNEL(k_buffer, "buffer_store_dwordx2 %0, %7, s[24:27], 0 offen\n s_add_u32 s24, s24, s20\n s_addc_u32 s25, s25, s21\n")

typedef void (*kern_t)(double *, unsigned long long *, double, long long);
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    struct { const char *name; kern_t fn; } tab[] = {
        {"no store (pointer add only)", k_none}, {"global_store_dwordx2 off", k_x2}, {"global_store_dwordx2 off nt", k_x2_nt},
        {"global_store_dwordx2 off sc1", k_x2_sc1}, {"global_store_dwordx2 off sc0 sc1", k_x2_sc0sc1}, {"global_store_dword off", k_x1},
        {"global_store_dwordx2 saddr + 2 salu", k_saddr}, {"buffer_store_dwordx2 offen + 2 salu", k_buffer}};
    const int blocks = 1024;
    // per store a wavefront touches 512 B at `stride` from its previous store; 256 stores per wavefront
    for (long long stride : {512LL * 1024, 64LL * 1024 * 8 /* 512 KB again */, 32768LL, 512LL}) {
        size_t bytes = (size_t)blocks * 512 + (size_t)260 * stride + (1 << 20);
        if (stride == 512) bytes = (size_t)blocks * 512 * 300;
        printf("---- stride between a wavefront's consecutive stores: %lld B\n", stride);
        for (auto &e : tab) {
            double *out; unsigned long long *cyc;
            hipMalloc(&out, bytes); hipMalloc(&cyc, blocks * sizeof(unsigned long long));
            long long st = stride;
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.5, st);
            hipError_t err = hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("  %-42s %8.1f cycles per (store + 128 fma) group %s\n", e.name, s / blocks / 256.0, err == hipSuccess ? "" : hipGetErrorString(err));
            hipFree(out); hipFree(cyc);
        }
    }
    return 0;
}

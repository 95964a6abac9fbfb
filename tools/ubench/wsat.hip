// Where does the streaming-store path saturate?  Every wavefront (one per SIMD, 1024 of them) issues 24 stores per 768 FMAs
// -- denser than the memory system can drain -- with different widths, cache modifiers and lane->address patterns.
// Reports the sustained unique bytes/s.  build: hipcc -O3 --offload-arch=gfx950 tools/ubench/wsat.hip -o tools/ubench/wsat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double2v __attribute__((ext_vector_type(2)));
#define ITERS 256
#define R4(x) x x x x
#define R12(x) R4(x) R4(x) R4(x)
#define FMA8 "v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n"
#define FMA32 R4(FMA8)
#define PADD "v_lshl_add_u64 %4, %4, 0, s[20:21]\n"
// operands: %0 d2 (4 VGPRs), %1-%3 accumulators, %4 per-lane pointer, %5 b, %6 c, %7 d1 (2 VGPRs), %8 d0 (1 VGPR)

// MODE 0: lanes 8 B apart (512 B per store); 1: lanes 16 B apart (1 KB); 2: the kernel's pattern, even lanes one 256 B segment,
// odd lanes another one `seg` bytes away; 3: lanes 4 B apart (256 B per store)
#define KERNEL(NAME, MODE, BODY)                                                                                   \
    __global__ __launch_bounds__(64) void NAME(char *out, double seed, long long stride, long long seg) {            \
        double2v d2 = {seed + threadIdx.x, seed - threadIdx.x};                                                     \
        double d1 = seed * threadIdx.x, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = 1.0000001, c = 1e-9;       \
        float d0 = (float)seed;                                                                                    \
        const unsigned l = threadIdx.x;                                                                            \
        char *gp = out + (MODE == 0 ? (size_t)blockIdx.x * 512 + l * 8                                             \
                        : MODE == 1 ? (size_t)blockIdx.x * 1024 + l * 16                                           \
                        : MODE == 2 ? (size_t)blockIdx.x * 256 + (l >> 1) * 8 + (l & 1) * seg                      \
                                    : (size_t)blockIdx.x * 256 + l * 4);                                           \
        asm volatile("s_mov_b32 s20, %0\n s_mov_b32 s21, %1\n" :: "s"((unsigned)stride), "s"((unsigned)(stride >> 32)) : "s20", "s21"); \
        for (int it = 0; it < ITERS; ++it) {                                                                       \
            asm volatile(R12(BODY PADD FMA32 BODY PADD FMA32) : "+v"(d2), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(gp) : "v"(b), "v"(c), "v"(d1), "v"(d0) : "memory"); \
        }                                                                                                          \
        if (a1 + a2 + a3 == 0.123) *(double *)out = a1;                                                            \
    }

KERNEL(k_x2, 0, "global_store_dwordx2 %4, %7, off\n")
KERNEL(k_x2_nt, 0, "global_store_dwordx2 %4, %7, off nt\n")
KERNEL(k_x2_sc, 0, "global_store_dwordx2 %4, %7, off sc0 sc1\n")
KERNEL(k_x4, 1, "global_store_dwordx4 %4, %0, off\n")
KERNEL(k_x4_nt, 1, "global_store_dwordx4 %4, %0, off nt\n")
KERNEL(k_x2_seg, 2, "global_store_dwordx2 %4, %7, off\n")
KERNEL(k_x1, 3, "global_store_dword %4, %8, off\n")

typedef void (*kern_t)(char *, double, long long, long long);
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int blocks = 1024;
    // bytes per store instruction and the stride that makes consecutive store steps of the grid tile memory without overlap
    struct { const char *name; kern_t fn; long long per_store, stride, seg; } tab[] = {
        {"dwordx2, 512 B contiguous per wavefront", k_x2, 512, 512LL * blocks, 0},
        {"dwordx2 nt", k_x2_nt, 512, 512LL * blocks, 0},
        {"dwordx2 sc0 sc1", k_x2_sc, 512, 512LL * blocks, 0},
        {"dwordx4, 1 KB contiguous per wavefront", k_x4, 1024, 1024LL * blocks, 0},
        {"dwordx4 nt", k_x4_nt, 1024, 1024LL * blocks, 0},
        {"dwordx2, 2 x 256 B segments (kernel's pattern)", k_x2_seg, 512, 256LL * blocks * 2, 256LL * blocks},
        {"dword, 256 B contiguous per wavefront", k_x1, 256, 256LL * blocks, 0}};
    for (auto &e : tab) {
        const size_t bytes = (size_t)(ITERS * 24 + 8) * e.stride + (1 << 20);
        char *out;
        if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e30f;
        for (int w = 0; w < 3; ++w) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, out, 1.5, e.stride, e.seg);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        hipError_t err = hipDeviceSynchronize();
        const double total = (double)blocks * ITERS * 24 * e.per_store;
        printf("  %-48s %7.3f ms  %6.2f GB  %6.2f TB/s %s\n", e.name, best, total / 1e9, total / best / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
        (void)hipFree(out);
    }
    return 0;
}

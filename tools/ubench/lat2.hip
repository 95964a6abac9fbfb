// Microbenchmark 2: what an fp64 VALU instruction costs a LONE wavefront depending on (a) its kind (fma / add / mul), (b) how many instructions
// ago its register operand was written (1, 2, 4, 8), (c) how many of its operands are vector registers.  One wavefront per SIMD (1024 blocks).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/lat2.hip -o <somewhere that travels> ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001 + 1e-12 * threadIdx.x, c = 1e-9 * (1 + threadIdx.x);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE == 0) asm volatile(R8("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));                       // dependent fma, 3 VGPR operands
            if (MODE == 1) asm volatile(R8("v_add_f64 %0, %0, %1\n") : "+v"(a0) : "v"(c));                                   // dependent add
            if (MODE == 2) asm volatile(R8("v_fma_f64 %0, %0, 1.0, %1\n") : "+v"(a0) : "v"(c));                              // dependent add written as fma
            if (MODE == 3) asm volatile(R8("v_mul_f64 %0, %0, %1\n") : "+v"(a0) : "v"(b));                                   // dependent mul
            if (MODE == 4) asm volatile(R8("v_fma_f64 %0, %0, %1, 0\n") : "+v"(a0) : "v"(b));                                // dependent mul written as fma (+0: timing only)
            if (MODE == 5) asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                                        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));                         // 4 chains fma
            if (MODE == 6) asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                                        "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));                                 // 4 chains add
            if (MODE == 7) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                                        "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                                        : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));                                             // 2 chains fma
            if (MODE == 8) asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                                        "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));   // 8 chains fma
            if (MODE == 9) asm volatile("v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n"
                                        "v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n"
                                        : "+v"(a0) : "v"(b), "v"(c));                                                         // dependent fmac (VOP2 encoding)
            if (MODE == 10) asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5\n"
                                         "v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));                        // 4 chains fmac
            if (MODE == 11) asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                                         "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));                                // 4 chains mul
            if (MODE == 12) asm volatile("v_fma_f64 %0, %0, 1.0, %4\n v_fma_f64 %1, %1, 1.0, %4\n v_fma_f64 %2, %2, 1.0, %4\n v_fma_f64 %3, %3, 1.0, %4\n"
                                         "v_fma_f64 %0, %0, 1.0, %4\n v_fma_f64 %1, %1, 1.0, %4\n v_fma_f64 %2, %2, 1.0, %4\n v_fma_f64 %3, %3, 1.0, %4\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));                                // 4 chains add-as-fma
            if (MODE == 13) asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n"
                                         : "+v"(((int *)&a0)[0]), "+v"(((int *)&a0)[1]));                                  // dependent v_mov_b32
            if (MODE == 14) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_read_b32 %1, a1\n"
                                         "v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_read_b32 %1, a1\n"
                                         : "+v"(((int *)&a0)[0]), "+v"(((int *)&a0)[1]) :: "a0", "a1");                       // accvgpr write + read pairs
            if (MODE == 15) asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n"
                                         "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n"
                                         : "+v"(((int *)&a0)[0]), "+v"(((int *)&a0)[1]), "+v"(((int *)&a1)[0]), "+v"(((int *)&a1)[1]));   // permlane32 swaps
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks) {
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, blocks * 64 * sizeof(double)); (void)hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.5);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-52s %5d wavefronts: %6.2f cycles/instruction\n", name, blocks, s / blocks / (64.0 * 256));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int blocks : {1024, 2048}) {
        run<0>("fma dependent (3 VGPR operands)", blocks); run<7>("fma 2 chains", blocks); run<5>("fma 4 chains", blocks); run<8>("fma 8 chains", blocks);
        run<9>("fmac dependent", blocks); run<10>("fmac 4 chains", blocks);
        run<1>("add dependent", blocks); run<6>("add 4 chains", blocks); run<2>("add as fma(a, 1.0, c) dependent", blocks); run<12>("add as fma, 4 chains", blocks);
        run<3>("mul dependent", blocks); run<11>("mul 4 chains", blocks); run<4>("mul as fma(a, b, 0) dependent", blocks);
        run<13>("v_mov_b32 dependent", blocks); run<14>("v_accvgpr write + read", blocks); run<15>("v_permlane32_swap", blocks);
    }
    return 0;
}

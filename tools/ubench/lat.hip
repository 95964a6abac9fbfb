// Microbenchmark: cycles per instruction for dependent vs independent chains, one wavefront per SIMD (1024 single-wave blocks).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/lat.hip -o tools/ubench/lat ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    __shared__ double lds[64 * 8];
    lds[threadIdx.x] = a0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int i = 0; i < REP / 8; ++i) {
            if (MODE == 0) {  // dependent fma chain (8 per iteration)
                a0 = fma(a0, b, c); a0 = fma(a0, b, c); a0 = fma(a0, b, c); a0 = fma(a0, b, c);
                a0 = fma(a0, b, c); a0 = fma(a0, b, c); a0 = fma(a0, b, c); a0 = fma(a0, b, c);
            } else if (MODE == 1) {  // 8 independent chains
                a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
                a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
            } else if (MODE == 2) {  // 2 independent chains
                a0 = fma(a0, b, c); a1 = fma(a1, b, c); a0 = fma(a0, b, c); a1 = fma(a1, b, c);
                a0 = fma(a0, b, c); a1 = fma(a1, b, c); a0 = fma(a0, b, c); a1 = fma(a1, b, c);
            } else if (MODE == 3) {  // dependent rcp
                a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0);
                a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0);
            } else if (MODE == 4) {  // independent rcp
                a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
                a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7);
            } else if (MODE == 5) {  // dependent add f64
                a0 += c; a0 += c; a0 += c; a0 += c; a0 += c; a0 += c; a0 += c; a0 += c;
            } else if (MODE == 6) {  // 4 independent chains
                a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
                a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
            } else if (MODE == 7) {  // dependent LDS read -> fma -> address (latency of ds_read_b64)
                a0 = fma(lds[((int)a0) & 63], 1e-30, 1.0); a0 = fma(lds[((int)a0) & 63], 1e-30, 1.0);
                a0 = fma(lds[((int)a0) & 63], 1e-30, 1.0); a0 = fma(lds[((int)a0) & 63], 1e-30, 1.0);
                a0 = fma(lds[((int)a0) & 63], 1e-30, 1.0); a0 = fma(lds[((int)a0) & 63], 1e-30, 1.0);
                a0 = fma(lds[((int)a0) & 63], 1e-30, 1.0); a0 = fma(lds[((int)a0) & 63], 1e-30, 1.0);
            } else if (MODE == 8) {  // dependent DPP swap + add (pair_sum)
                for (int u = 0; u < 8; ++u) {
                    int lo = __double2loint(a0), hi = __double2hiint(a0);
                    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true);
                    a0 = a0 * 0.5 + __hiloint2double(hi, lo) * 0.5;
                }
            } else if (MODE == 9) {  // independent 32-bit moves/adds (int)
                int *p = (int *)&a0; 
                asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1" : "+v"(p[0]), "+v"(p[1]));
            } else if (MODE == 10) {  // independent SALU
                int s0 = it, s1 = i;
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1" : "+s"(s0), "+s"(s1));
                a1 += s0 + s1;
            } else if (MODE == 11) {  // alternating independent fma f64 and SALU
                int s0 = it;
                asm volatile("v_fma_f64 %0, %0, %4, %5\n s_add_u32 %3, %3, 1\n v_fma_f64 %1, %1, %4, %5\n s_add_u32 %3, %3, 1\n v_fma_f64 %2, %2, %4, %5\n s_add_u32 %3, %3, 1\n v_fma_f64 %0, %0, %4, %5\n s_add_u32 %3, %3, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+s"(s0) : "v"(b), "v"(c));
                a3 += s0;
            } else if (MODE == 12) {  // dependent mul f64
                a0 *= b; a0 *= b; a0 *= b; a0 *= b; a0 *= b; a0 *= b; a0 *= b; a0 *= b;
            } else if (MODE == 13) {  // independent v_mov_b64 / cndmask style: 8 independent v_cndmask_b32
                int *p = (int *)&a0; int *q2 = (int *)&a1;
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(p[0]), "+v"(p[1]), "+v"(q2[0]), "+v"(q2[1]) :: "vcc");
            }
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
    hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.5);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-44s %2d blocks/SIMD-ish(%5d): %6.2f cycles/op\n", name, blocks / 1024, blocks, s / blocks / (64.0 * REP));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1024, 2048}) {
        run<0>("fma f64 dependent", blocks); run<2>("fma f64 2 chains", blocks); run<6>("fma f64 4 chains", blocks); run<1>("fma f64 8 chains", blocks);
        run<5>("add f64 dependent", blocks); run<12>("mul f64 dependent", blocks);
        run<3>("rcp f64 dependent", blocks); run<4>("rcp f64 8 independent", blocks);
        run<7>("lds read->fma->addr dependent (per pair)", blocks); run<8>("dpp swap(2 mov)+2mul-add dependent (per group)", blocks);
        run<9>("v_add_u32 2 chains", blocks); run<10>("s_add_u32 2 chains", blocks); run<11>("fma f64 + s_add alternating (per pair... /2)", blocks);
        run<13>("v_cndmask_b32 (4 regs)", blocks);
    }
    return 0;
}

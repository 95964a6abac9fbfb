// Which SIMD does each wavefront of a 512-thread workgroup land on?  (HW_REG_HW_ID: SIMD_ID bits 5:4, CU_ID 11:8 ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned *out) {
    __shared__ double pad[18000];                      // ~144 KB of LDS: one workgroup per CU
    pad[threadIdx.x] = 1.0;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = id;
    __syncthreads();
    if (pad[threadIdx.x] == 2.0) out[0] = 0;
}
int main() {
    unsigned *d; hipMalloc(&d, 512 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d);
    unsigned h[512 * 8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int hist[4][4] = {};  // [wave%4][simd]
    int same = 0, total = 0;
    for (int b = 0; b < 512; ++b) {
        for (int w = 0; w < 8; ++w) { unsigned simd = (h[b * 8 + w] >> 4) & 3; if (b < 6) printf("%u ", simd); hist[w % 4][simd]++; }
        if (b < 6) printf("\n");
        for (int w = 0; w < 4; ++w) { same += (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3)); ++total; }
    }
    printf("waves w and w+4 on the same SIMD: %d of %d\n", same, total);
    return 0;
}

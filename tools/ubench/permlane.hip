// What v_permlane32_swap does on gfx950: r = __builtin_amdgcn_permlane32_swap(a, b, false, false) with a = lane, b = 100 + lane.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/permlane.hip -o tools/ubench/permlane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned a = threadIdx.x, b = 100u + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    (void)hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int part = 0; part < 2; ++part) {
        printf("r[%d]:", part);
        for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[part * 64 + i]);
        printf("  [31]=%u [32]=%u [63]=%u\n", h[part * 64 + 31], h[part * 64 + 32], h[part * 64 + 63]);
    }
    return 0;
}

// Which lane does each DPP control code used by the kernels read from?  Every lane moves its own lane id through the code and the host
// checks the permutation against what rmckf_device.hpp / rmckf_tuned.hpp assume (quad_perm butterflies, row_half_mirror, row_mirror,
// row_newbcast).  build: hipcc -O3 --offload-arch=gfx950 tools/ubench/dpp_test.hip -o tools/ubench/dpp_test
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ int mov(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }

__global__ void k(int *o) {
    const int l = (int)threadIdx.x;
    o[0 * 64 + l] = mov<0xB1>(l);                    // quad_perm [1,0,3,2]: lane ^ 1
    o[1 * 64 + l] = mov<0x4E>(l);                    // quad_perm [2,3,0,1]: lane ^ 2
    o[2 * 64 + l] = mov<0xA0>(l);                    // quad_perm [0,0,2,2]: even lane of the pair
    o[3 * 64 + l] = mov<0xF5>(l);                    // quad_perm [1,1,3,3]: odd lane of the pair
    o[4 * 64 + l] = mov<0x141>(l);                   // row_half_mirror: 7 - (lane & 7) within each half row
    o[5 * 64 + l] = mov<0x140>(l);                   // row_mirror: 15 - (lane & 15) within each row
    o[6 * 64 + l] = mov<0x150 + 5>(l);               // row_newbcast:5: lane 5 of the row
    o[7 * 64 + l] = mov<0x55 * 3>(l);                // quad_perm [3,3,3,3]
    // two groups of 8 in one row (rmckf_device.hpp group_bcast32<8, OWNER>): bank masks select the half rows
    int r = __builtin_amdgcn_update_dpp(0, l, 0x150 + 2, 0xf, 0x3, false);
    o[8 * 64 + l] = __builtin_amdgcn_update_dpp(r, l, 0x150 + 8 + 2, 0xf, 0xC, false);
}

int main() {
    int *d, h[9 * 64];
    (void)hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l & ~15, half = l & ~7, quad = l & ~3;
        const int want[9] = {l ^ 1, l ^ 2, l & ~1, l | 1, half + 7 - (l & 7), row + 15 - (l & 15), row + 5, quad + 3, half + 2};
        for (int c = 0; c < 9; ++c)
            if (h[c * 64 + l] != want[c]) { ++bad; printf("code %d lane %d: got %d want %d\n", c, l, h[c * 64 + l], want[c]); }
    }
    printf(bad ? "DPP control codes: %d mismatches\n" : "DPP control codes: all 9 x 64 as assumed (%d mismatches)\n", bad);
    return bad != 0;
}

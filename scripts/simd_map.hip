// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_ID: wave 3:0, SIMD 5:4, CU 11:8)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/simd_map scripts/simd_map.hip && scripts/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4 * 8 * 4);
    k<<<4, 512, 100 * 1024>>>(d);
    unsigned h[32];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; b++) {
        printf("workgroup %d:", b);
        for (int w = 0; w < 8; w++) printf("  w%d simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}

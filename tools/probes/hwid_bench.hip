// Which SIMD does wave w of a 768-thread workgroup run on?  (HW_REG_HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8], se_id[15:13])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(768) void k(unsigned *out) {
    extern __shared__ char dyn[];
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = v;
    if (threadIdx.x == 5000) dyn[0] = 1;
}
int main() {
    unsigned *d, h[12 * 8];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(8), dim3(768), 150 * 1024, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 8; b++) {
        printf("wg %d: simd of waves 0..11:", b);
        for (int w = 0; w < 12; w++) printf(" %u", (h[b * 12 + w] >> 4) & 3);
        printf("   cu %u se %u\n", (h[b * 12] >> 8) & 15, (h[b * 12] >> 13) & 7);
    }
    return 0;
}

// MFMA latency / issue microbenchmark on gfx950: cycles per instruction for a dependent chain
// and for 3 / 4 independent chains, one wave per SIMD (block of 256) or a single wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k_f64(double *out, unsigned long long *cyc, int n) {
    f64x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = {0, 0, 0, 0};
    double a = threadIdx.x * 0.5, b = 1.0 + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS>
__global__ void k_f32(float *out, unsigned long long *cyc, int n) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = {0, 0, 0, 0};
    float a = threadIdx.x * 0.5f, b = 1.0f + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS>
__global__ void k_f64_4x4(double *out, unsigned long long *cyc, int n) {   // v_mfma_f64_4x4x4_4b_f64: four 4x4x4 blocks per instruction
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = 0;
    double a = threadIdx.x * 0.5, b = 1.0 + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k_dfma(double *out, unsigned long long *cyc, int n) {
    double acc[4] = {0, 0, 0, 0};
    double a = threadIdx.x * 0.5, b = 1.0 + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] = fma(a, b, acc[c]);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double *o64; float *o32; unsigned long long *cyc, h;
    hipMalloc(&o64, 1 << 20); hipMalloc(&o32, 1 << 20); hipMalloc(&cyc, 8);
    const int n = 1000;
#define RUN(name, kern, buf, threads, per)                                             \
    kern<<<1, threads>>>(buf, cyc, n); kern<<<1, threads>>>(buf, cyc, n);              \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                      \
    printf("%-44s %7.1f ticks per instruction\n", name, (double)h / (n * per));
    RUN("f64 mfma 16x16x4, 1 chain, 1 wave", k_f64<1>, o64, 64, 1)
    RUN("f64 mfma 16x16x4, 3 chains, 1 wave", k_f64<3>, o64, 64, 3)
    RUN("f64 mfma 16x16x4, 4 chains, 1 wave", k_f64<4>, o64, 64, 4)
    RUN("f64 mfma 16x16x4, 4 chains, 4 waves", k_f64<4>, o64, 256, 4)
    RUN("f64 mfma 4x4x4 (4 blocks), 1 chain, 1 wave", k_f64_4x4<1>, o64, 64, 1)
    RUN("f64 mfma 4x4x4 (4 blocks), 4 chains, 1 wave", k_f64_4x4<4>, o64, 64, 4)
    RUN("f64 mfma 4x4x4 (4 blocks), 8 chains, 1 wave", k_f64_4x4<8>, o64, 64, 8)
    RUN("f64 mfma 4x4x4 (4 blocks), 8 chains, 4 waves", k_f64_4x4<8>, o64, 256, 8)
    RUN("f32 mfma 16x16x4, 1 chain, 1 wave", k_f32<1>, o32, 64, 1)
    RUN("f32 mfma 16x16x4, 2 chains, 1 wave", k_f32<2>, o32, 64, 2)
    RUN("f32 mfma 16x16x4, 4 chains, 1 wave", k_f32<4>, o32, 64, 4)
    RUN("f32 mfma 16x16x4, 4 chains, 8 waves", k_f32<4>, o32, 512, 4)
    RUN("v_fma_f64, 4 chains, 1 wave", k_dfma, o64, 64, 4)
    RUN("v_fma_f64, 4 chains, 8 waves", k_dfma, o64, 512, 4)
    return 0;
}

// Does an MFMA-spinning wave slow a VALU wave on the same SIMD?  8 waves per block (2 per SIMD):
// waves 0-3 run the "partner" loop (idle / f32 MFMA / f64 MFMA / LDS reads), waves 4-7 run a VALU + LDS-read
// loop that mimics the consumer; the consumer waves' cycles are reported.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mode, int n, float *out, unsigned long long *cyc) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 0.001f;
    __syncthreads();
    if (wave < 4) {
        if (mode == 1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            float x = lane * 0.5f, y = lane + 1.f;
            for (int i = 0; i < n * 4; i++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            }
            out[threadIdx.x] = a0[0] + a1[1];
        } else if (mode == 2) {
            f64x4 a0 = {0, 0, 0, 0};
            double x = lane * 0.5, y = lane + 1.0;
            for (int i = 0; i < n * 4; i++) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            out[threadIdx.x] = (float)a0[0];
        } else if (mode == 3) {
            float s = 0;
            for (int i = 0; i < n * 16; i++) s += lds[(lane * 4 + i * 64) & 8191];
            out[threadIdx.x] = s;
        }
    } else {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; i++) {
            const f32x4 t = *(const f32x4 *)&lds[((lane * 4) + i * 256) & 8188];
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    acc[e] = fmaf(t[e], 1.0001f, acc[e]);
                    acc[4 + e] = fmaf(t[e], 0.9999f, acc[4 + e]);
                }
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int e = 0; e < 8; e++) s += acc[e];
        out[threadIdx.x] = s;
        if (lane == 0) cyc[wave - 4] = t1 - t0;
    }
}
int main() {
    float *out; unsigned long long *cyc, h[4];
    hipMalloc(&out, 4096); hipMalloc(&cyc, 32);
    const char *names[] = {"partner idle", "partner f32 MFMA 16x16x4 (2 chains)", "partner f64 MFMA 16x16x4", "partner LDS reads"};
    const int n = 2000;
    for (int mode = 0; mode < 4; mode++) {
        k<<<1, 512>>>(mode, n, out, cyc); k<<<1, 512>>>(mode, n, out, cyc);
        hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        printf("%-40s consumer-wave ticks per iteration (32 fma + 1 ds_read_b128): %.1f %.1f %.1f %.1f\n", names[mode],
               (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, (double)h[3] / n);
    }
    return 0;
}

// Store-pattern replica for "the sweep's taps on the matrix pipe" (DESIGN 4.1 round 4 (b), 8.1): what would the store stream do
// if a store instruction wrote the D tile of v_mfma_f32_16x16x4 -- lane l: 4 consecutive channels 16 mt + 4 (l / 16) .. + 3 of sample
// 16 nt + l % 16, i.e. 16 runs of 64 bytes 316 bytes apart -- instead of today's three whole rows (948 consecutive bytes)?
// Same unit -> workgroup map, same bytes, no production; a plain fill beside them.  usage: store_pattern_mfma [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define T 156
#define D 79
#define NF 39
#define NCH 4
typedef float f32x4pu __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3pu __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x4p __attribute__((ext_vector_type(4)));

// MODE 0: today's pattern (three rows per instruction); 1: D-tile pattern, for nt: for mt; 2: D-tile pattern, for mt: for nt;
// 3: D-tile pattern with the two candidates of a wave interleaved per M tile
template <int MODE>
__global__ __launch_bounds__(512) void pattern_kernel(float *out, int ntiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int U = ntiles * NCH, per = (U + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int s = 0; s < per; s++) {
        const int u = blockIdx.x * per + s;
        if (u >= U) break;
        const int tile = u / NCH, chunk = (u % NCH + blockIdx.x) % NCH;
        if (MODE == 0) {
            const int fsub = lane / 20, ql = lane % 20;
            for (int f0 = 0; f0 < NF; f0 += 3)
                for (int half = 0; half < 2; half++) {
                    const size_t cand = (size_t)tile * 16 + wave + 8 * half;
                    const int f = f0 + fsub;
                    if (lane < 60 && f < NF) {
                        float *p = out + (cand * T + (size_t)(chunk * NF + f)) * D + (ql == 19 ? 75 : 4 * ql);
                        const f32x4pu v = {0.f, 0.f, 0.f, 0.f};
                        *(f32x4pu *)p = v;
                    }
                }
        } else {
            const int n = lane & 15, q = lane >> 4;
            auto store = [&](int half, int nt, int mt) {
                const size_t cand = (size_t)tile * 16 + wave + 8 * half;
                const int f = 16 * nt + n, ch = 16 * mt + 4 * q;
                if (f < NF) {
                    float *p = out + (cand * T + (size_t)(chunk * NF + f)) * D + ch;
                    if (ch + 4 <= D) { const f32x4pu v = {0.f, 0.f, 0.f, 0.f}; *(f32x4pu *)p = v; }
                    else { const f32x3pu v = {0.f, 0.f, 0.f}; *(f32x3pu *)p = v; }
                }
            };
            if (MODE == 1) { for (int half = 0; half < 2; half++) for (int nt = 0; nt < 3; nt++) for (int mt = 0; mt < 5; mt++) store(half, nt, mt); }
            if (MODE == 2) { for (int half = 0; half < 2; half++) for (int mt = 0; mt < 5; mt++) for (int nt = 0; nt < 3; nt++) store(half, nt, mt); }
            if (MODE == 3) { for (int nt = 0; nt < 3; nt++) for (int mt = 0; mt < 5; mt++) for (int half = 0; half < 2; half++) store(half, nt, mt); }
        }
    }
}
__global__ __launch_bounds__(256) void fill_kernel(f32x4p *buf, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const f32x4p v = {0.f, 0.f, 0.f, 0.f}; buf[i] = v; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int B = 8192, ntiles = B / 16;
    const size_t bytes = (size_t)B * T * D * 4;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    // several buffers: the placement class of an allocation decides more than the pattern; report each
    for (int bi = 0; bi < 4; bi++) {
        float *buf; CK(hipMalloc(&buf, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time = [&](auto launch) {
            for (int i = 0; i < 3; i++) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < reps; i++) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); return 1e3 * ms / reps;
        };
        const size_t n4 = bytes / 16;
        double r[6];
        for (int round = 0; round < 2; round++) {
            r[0] = time([&] { hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (f32x4p *)buf, n4); });
            r[1] = time([&] { hipLaunchKernelGGL(pattern_kernel<0>, dim3(grid), dim3(512), 0, 0, buf, ntiles); });
            r[2] = time([&] { hipLaunchKernelGGL(pattern_kernel<1>, dim3(grid), dim3(512), 0, 0, buf, ntiles); });
            r[3] = time([&] { hipLaunchKernelGGL(pattern_kernel<2>, dim3(grid), dim3(512), 0, 0, buf, ntiles); });
            r[4] = time([&] { hipLaunchKernelGGL(pattern_kernel<3>, dim3(grid), dim3(512), 0, 0, buf, ntiles); });
            printf("buffer %d round %d: fill %.1f us | rows (today) %.1f | D tile nt-mt %.1f | D tile mt-nt %.1f | D tile, candidates interleaved %.1f\n", bi, round, r[0], r[1], r[2], r[3], r[4]);
        }
        CK(hipGetLastError());
        // leave the buffer allocated so that the next one lands elsewhere
    }
    return 0;
}

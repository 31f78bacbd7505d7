// Do the partially written 128-byte lines at time-chunk boundaries cost anything?  The sweep's geometry (256 persistent
// workgroups, 8 storing waves, wave = 2 candidates, flat float4 runs) with NCH chunks per candidate; a chunk's byte
// range per candidate is either the true one (rows f0 .. f1 of 316 bytes: boundaries fall inside lines, the line is
// written half by this unit and half by the next one ~20 K cycles later) or rounded to 128-byte lines (every line is
// written whole, once).  Build: hipcc --offload-arch=gfx950 -O3 -o store_bench8 store_bench8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int B = 8192, T = 156, D = 79, NTILES = B / 16;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));

__global__ __launch_bounds__(768) void k(float *out, int nch, int aligned) {
    extern __shared__ float dyn[];
    if (nch < 0) dyn[threadIdx.x] = 1.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) return;
    const int cj = wave - 4;
    const int G = gridDim.x, U = NTILES * nch, w = blockIdx.x, per = (U + G - 1) / G;
    for (int s = 0; s < per; s++) {
        const int u = w * per + s;
        if (u >= U) continue;
        const int tile = u / nch, chunk = u % nch;
        const int f0 = (int)((long)T * chunk / nch), f1 = (int)((long)T * (chunk + 1) / nch);
        for (int half = 0; half < 2; half++) {
            const size_t cand0 = (size_t)(tile * 16 + cj + 8 * half) * T * D;     // floats
            size_t e0 = cand0 + (size_t)f0 * D, e1 = cand0 + (size_t)f1 * D;
            if (aligned) {   // whole 128-byte lines (32 floats); the candidate's first and last line stay with the first / last chunk
                if (chunk > 0) e0 = (e0 + 31) / 32 * 32;
                if (chunk < nch - 1) e1 = (e1 + 31) / 32 * 32;
            }
            for (size_t i = e0 + 4 * lane; i + 4 <= e1; i += 256) {
                f4u v = {1.f, 2.f, 3.f, 4.f};
                *(f4u *)(out + i) = v;
            }
        }
    }
}

int main() {
    float *out;
    const size_t N = (size_t)B * T * D;
    CK(hipMalloc(&out, N * 4 + 4096));
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](std::function<void()> f) {
        for (int i = 0; i < 3; i++) f();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; i++) f();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 20;
    };
    for (int rep = 0; rep < 2; rep++)
        for (int nch : {1, 2, 3, 4, 6, 8, 12})
            for (int al = 0; al < 2; al++) {
                float ms = timeit([&] { k<<<256, 768, 150 * 1024>>>(out, nch, al); });
                printf("%2d chunks per candidate, %-22s %6.1f us  %7.1f GB/s\n", nch, al ? "line-aligned pieces" : "true row boundaries", ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
                fflush(stdout);
            }
    return 0;
}

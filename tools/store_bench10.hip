// The 16 candidates a workgroup writes at the same time are 49 296 bytes apart (12.04 pages of 4 KiB): do they collide
// on memory channels?  The sweep's store pattern (256 persistent workgroups, 8 storing waves, wave = 2 candidates,
// 952-byte quad-row pieces, 4 chunks of 39 rows) with the tile's 16 candidates spaced S candidates apart instead
// of being neighbours: tile t of a block of 16 S candidates holds candidates (t % S) + S k, k = 0..15.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bench10 store_bench10.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int B = 8192, T = 156, D = 79, NF = 39, NCH = 4, NTILES = B / 16;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));

__global__ __launch_bounds__(768) void k(float *out, int S) {
    extern __shared__ float dyn[];
    if (S < 0) dyn[threadIdx.x] = 1.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) return;
    const int cj = wave - 4;
    const int G = gridDim.x, U = NTILES * NCH, w = blockIdx.x, per = (U + G - 1) / G;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    for (int s = 0; s < per; s++) {
        const int u = w * per + s;
        if (u >= U) continue;
        const int tile = u / NCH, chunk = u % NCH;
        const int blk = tile / S, tin = tile % S;          // block of 16 S candidates, tile inside it
        for (int f0 = 0; f0 < NF; f0 += 3)
            for (int half = 0; half < 2; half++) {
                const int k16 = cj + 8 * half;
                const size_t cand = (size_t)blk * 16 * S + tin + (size_t)S * k16;
                const int f = f0 + fsub;
                if (on && f < NF) {
                    float *p = out + (cand * T + chunk * NF + f) * D + (ql == 19 ? 75 : 4 * ql);
                    f4u v = {1.f, 2.f, 3.f, (float)f};
                    *(f4u *)p = v;
                }
            }
    }
}

int main() {
    const size_t N = (size_t)B * T * D;
    float *bufs[6];
    for (int i = 0; i < 6; i++) CK(hipMalloc(&bufs[i], N * 4 + 4096));
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
    timeit([&] { k<<<256, 768, 150 * 1024>>>(bufs[0], 1); });
    printf("spacing S     :"); for (int S : {1, 2, 4, 8, 16, 32, 64, 128, 512}) printf(" %6d", S); printf("\n");
    for (int b = 0; b < 6; b++) {
        printf("buffer %d (us) :", b);
        for (int S : {1, 2, 4, 8, 16, 32, 64, 128, 512}) printf(" %6.1f", 1e3 * timeit([&] { k<<<256, 768, 150 * 1024>>>(bufs[b], S); }));
        printf("\n"); fflush(stdout);
    }
    return 0;
}

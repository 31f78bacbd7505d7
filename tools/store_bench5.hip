// In-workgroup store pattern x unit order, in the frames kernel's geometry (256 persistent workgroups of 768 threads,
// waves 4..11 store; a unit = 16 candidates x one 26-frame chunk = 16 pieces of 8216 bytes, 49296 bytes apart).
//   pattern 0: today's sweep -- wave j owns candidates j and j + 8, 948-byte quad-row pieces, two candidates in flight
//   pattern 1: candidate after candidate, the W active waves write the piece together, 1 KB (float4 / lane) per wave visit
//   pattern 2: like 1 but two candidates at a time (waves split 4 + 4)
//   pattern 3: wave j owns candidates j, j + 8 but writes flat float4 runs (1 KB per instruction) instead of quad rows
//   order 0: blocked (workgroup owns whole tiles)      order 1: unit s G + w (six neighbouring workgroups share a tile)
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bench5 store_bench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int B = 8192, T = 156, D = 79, NF = 26, NCH = 6, NTILES = B / 16;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef f3 f3u __attribute__((aligned(4)));

__device__ __forceinline__ void flat_run(float *out, size_t e0, size_t e1, int first, int step, int lane) {
    // floats [e0, e1) as unaligned float4 per lane; wave visits first, first + step, ... (1 KB each)
    for (size_t i = e0 + (size_t)first * 256 + 4 * lane; i + 4 <= e1; i += (size_t)step * 256) {
        f4u v = {1.f, 2.f, 3.f, 4.f};
        *(f4u *)(out + i) = v;
    }
}

__global__ __launch_bounds__(768) void k(float *out, int pattern, int order, int nwaves, int reverse = 0) {
    extern __shared__ float dyn[];
    if (pattern < 0) dyn[threadIdx.x] = 1.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4 || wave >= 4 + nwaves) return;
    const int cj = wave - 4;
    const int G = gridDim.x, U = NTILES * NCH, w = blockIdx.x, per = (U + G - 1) / G;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    const int d0 = 4 * ql, nst = ql == 19 ? 3 : 4;
    for (int s = 0; s < per; s++) {
        int u = order == 0 ? w * per + s : s * G + w;
        if (u >= U) continue;
        if (reverse) u = U - 1 - u;   // the whole walk backwards: what the previous launch wrote last is written first
        const int tile = u / NCH, chunk = u % NCH;
        auto piece = [&](int cand) { return ((size_t)(tile * 16 + cand) * T + chunk * NF) * D; };
        if (pattern == 0) {
            for (int f0 = 0; f0 < NF; f0 += 3)
                for (int half = 0; half < 2; half++) {
                    const int f = f0 + fsub;
                    if (on && f < NF) {
                        float *p = out + piece(cj + 8 * half) + (size_t)f * D + d0;
                        if (nst == 4) { f4u v = {1.f, 2.f, 3.f, (float)f}; *(f4u *)p = v; }
                        else { f3u v = {1.f, 2.f, 3.f}; *(f3u *)p = v; }
                    }
                }
        } else if (pattern == 4 || pattern == 5) {
            // all 16 candidates by `nwaves` waves: wave j owns candidates j, j + nwaves, ...; pattern 4: all of a wave's
            // candidates in flight per trip, pattern 5: two at a time
            const int nc = 16 / nwaves;
            if (pattern == 4) {
                for (int f0 = 0; f0 < NF; f0 += 3)
                    for (int q = 0; q < nc; q++) {
                        const int f = f0 + fsub;
                        if (on && f < NF) {
                            float *p = out + piece(cj + nwaves * q) + (size_t)f * D + d0;
                            if (nst == 4) { f4u v = {1.f, 2.f, 3.f, (float)f}; *(f4u *)p = v; }
                            else { f3u v = {1.f, 2.f, 3.f}; *(f3u *)p = v; }
                        }
                    }
            } else {
                for (int q0 = 0; q0 < nc; q0 += 2)
                    for (int f0 = 0; f0 < NF; f0 += 3)
                        for (int q = q0; q < q0 + 2 && q < nc; q++) {
                            const int f = f0 + fsub;
                            if (on && f < NF) {
                                float *p = out + piece(cj + nwaves * q) + (size_t)f * D + d0;
                                if (nst == 4) { f4u v = {1.f, 2.f, 3.f, (float)f}; *(f4u *)p = v; }
                                else { f3u v = {1.f, 2.f, 3.f}; *(f3u *)p = v; }
                            }
                        }
            }
        } else if (pattern == 1) {
            for (int cand = 0; cand < 16; cand++) flat_run(out, piece(cand), piece(cand) + (size_t)NF * D, cj, nwaves, lane);
        } else if (pattern == 2) {
            const int h = nwaves / 2, grp = cj / h, wi = cj % h;
            for (int cand = grp; cand < 16; cand += 2) flat_run(out, piece(cand), piece(cand) + (size_t)NF * D, wi, h, lane);
        } else {
            for (int half = 0; half < 2; half++) flat_run(out, piece(cj + 8 * half), piece(cj + 8 * half) + (size_t)NF * D, 0, 1, lane);
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
        for (int pattern = 4; pattern <= 5; pattern++)
            for (int nw : {1, 2, 4, 8})
                for (int lds : {150 * 1024, 0}) {
                    float ms = timeit([&] { k<<<256, 768, lds>>>(out, pattern, 0, nw); });
                    printf("all 16 candidates by %d waves, %s  lds %3dK  %6.1f us  %7.1f GB/s\n", nw, pattern == 4 ? "all in flight " : "two at a time", lds / 1024, ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
                    fflush(stdout);
                }
    // Infinity Cache (256 MB) as a write-back cache between launches: if every other launch walks the buffer backwards,
    // the last ~256 MB of launch i are the first of launch i + 1 and may be overwritten in the cache
    for (int rep = 0; rep < 3; rep++) {
        int flip = 0;
        float ms = timeit([&] { k<<<256, 768, 150 * 1024>>>(out, 0, 0, 8, 0); });
        printf("today's sweep, same direction every launch        %6.1f us  %7.1f GB/s\n", ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
        ms = timeit([&] { k<<<256, 768, 150 * 1024>>>(out, 0, 0, 8, flip); flip ^= 1; });
        printf("today's sweep, direction alternates per launch    %6.1f us  %7.1f GB/s\n", ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
        fflush(stdout);
    }
    const char *pn[] = {"quad rows, wave = 2 candidates (today)", "waves share a candidate, one at a time", "waves share, two candidates at a time", "flat float4 runs, wave = 2 candidates"};
    for (int rep = 0; rep < 2; rep++)
        for (int pattern = 0; pattern < 4; pattern++)
            for (int order = 0; order < 2; order++)
                for (int nw : {8}) {
                    float ms = timeit([&] { k<<<256, 768, 150 * 1024>>>(out, pattern, order, nw); });
                    printf("%-42s %-10s %d waves  %6.1f us  %7.1f GB/s\n", pn[pattern], order ? "joint" : "blocked", nw, ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
                    fflush(stdout);
                }
    return 0;
}

// Can the fast placement be MADE?  The output buffer is assembled with the virtual-memory API from physical chunks of a
// chosen size (hipMemCreate per chunk, mapped back to back), and the sweep's store pattern (store_bench10.hip) is timed
// on it; plain hipMalloc buffers for comparison.  Build: hipcc --offload-arch=gfx950 -O3 -o store_bench11 store_bench11.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int B = 8192, T = 156, D = 79, NF = 39, NCH = 4, NTILES = B / 16;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));

__device__ int g_rot = 0;   // 1: each workgroup walks a chunk's rows from its own starting row (de-synchronised offsets)
__global__ __launch_bounds__(768) void k(float *out, int ntiles = NTILES) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) return;
    const int cj = wave - 4;
    const int G = gridDim.x, U = ntiles * NCH, w = blockIdx.x, per = (U + G - 1) / G;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    for (int s = 0; s < per; s++) {
        const int u = w * per + s;
        if (u >= U) continue;
        const int tile = u / NCH, chunk = u % NCH;
        const int r0 = g_rot ? 3 * (int)((blockIdx.x * 5u + (unsigned)s * 3u) % 13u) : 0;
        for (int fi = 0; fi < NF; fi += 3)
            for (int half = 0; half < 2; half++) {
                const size_t cand = (size_t)tile * 16 + cj + 8 * half;
                const int f0 = (fi + r0) % NF;
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
    const size_t NB = (size_t)B * T * D * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](float *buf) {
        for (int i = 0; i < 3; i++) k<<<256, 768, 150 * 1024>>>(buf);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; i++) k<<<256, 768, 150 * 1024>>>(buf);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3f / 20;
    };
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float *warm; CK(hipMalloc(&warm, NB)); timeit(warm); timeit(warm);
    printf("hipMalloc, exact size :");
    std::vector<float *> plain;
    for (int i = 0; i < 6; i++) { float *p; CK(hipMalloc(&p, NB)); plain.push_back(p); printf(" %6.1f", timeit(p)); }
    printf("  us\n");
    { int one = 1, zero = 0;
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rot), &one, sizeof(int)));
      printf("  rotated row order   :");
      for (float *p : plain) printf(" %6.1f", timeit(p));
      printf("  us\n");
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rot), &zero, sizeof(int))); }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("minimum granularity %zu bytes\n", gran);
    for (size_t chunk : {(size_t)2 << 20, (size_t)8 << 20, (size_t)64 << 20, (size_t)512 << 20}) {
        if (chunk < gran) continue;
        printf("VMM, chunks of %4zu MiB:", chunk >> 20);
        for (int rep = 0; rep < 4; rep++) {
            const size_t n = (NB + chunk - 1) / chunk, total = n * chunk;
            void *va = nullptr;
            CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
            std::vector<hipMemGenericAllocationHandle_t> hs(n);
            for (size_t i = 0; i < n; i++) {
                CK(hipMemCreate(&hs[i], chunk, &prop, 0));
                CK(hipMemMap((char *)va + i * chunk, chunk, 0, hs[i], 0));
            }
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, total, &acc, 1));
            printf(" %6.1f", timeit((float *)va));
            fflush(stdout);
            // keep it mapped: later buffers must come from other physical memory
        }
        printf("  us\n");
    }
    // classify 64 MiB physical chunks one by one with the same pattern on 1360 candidates (85 tiles), then build a
    // full output from the fastest seven and time the real pattern on it
    {
        const size_t chunk = (size_t)64 << 20;
        const int NCHUNK = 28, TL = (int)(chunk / ((size_t)16 * T * D * 4));   // 85 tiles fit one chunk
        std::vector<hipMemGenericAllocationHandle_t> hs(NCHUNK);
        std::vector<float> tus(NCHUNK);
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        void *va7 = nullptr;
        CK(hipMemAddressReserve(&va7, 7 * chunk, 0, nullptr, 0));
        printf("each 64 MiB chunk mapped seven times in a row, full pattern (us):");
        for (int c = 0; c < NCHUNK; c++) {
            CK(hipMemCreate(&hs[c], chunk, &prop, 0));
            for (int i = 0; i < 7; i++) CK(hipMemMap((char *)va7 + i * chunk, chunk, 0, hs[c], 0));
            CK(hipMemSetAccess(va7, 7 * chunk, &acc, 1));
            tus[c] = timeit((float *)va7);
            printf(" %.1f", tus[c]);
            fflush(stdout);
            CK(hipMemUnmap(va7, 7 * chunk));
        }
        printf("\n");
        (void)TL;
        std::vector<int> order(NCHUNK);
        for (int c = 0; c < NCHUNK; c++) order[c] = c;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return tus[x] < tus[y]; });
        for (int pick = 0; pick < 2; pick++) {   // fastest seven, slowest seven
            void *va = nullptr;
            CK(hipMemAddressReserve(&va, 7 * chunk, 0, nullptr, 0));
            for (int i = 0; i < 7; i++) CK(hipMemMap((char *)va + i * chunk, chunk, 0, hs[order[pick == 0 ? i : NCHUNK - 1 - i]], 0));
            CK(hipMemSetAccess(va, 7 * chunk, &acc, 1));
            printf("full pattern on the %s seven chunks: %.1f %.1f us\n", pick == 0 ? "fastest" : "slowest", timeit((float *)va), timeit((float *)va));
            CK(hipMemUnmap(va, 7 * chunk));
        }
    }
    return 0;
}

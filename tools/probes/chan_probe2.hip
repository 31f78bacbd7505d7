// Follow-up to chan_probe.hip: what decides whether the sweep's store pattern (4096 concurrent ~1 KB-piece streams,
// 49 296 bytes apart) runs in the fast class (~63 us, the fill rate) or the slow one (~79 us)?
//  E1  base offset inside ONE physically contiguous 1 GiB chunk (2 MiB steps, then finer steps)
//  E2  two buffers written alternately (no cache carry-over from launch to launch) against one buffer rewritten
//  E5  many exact-size hipMalloc buffers held at once: which are fast (a map of the card's memory)?
//  E6  the same pattern with another candidate stride (diagnostic only: the product's layout is dense)
// Build: hipcc --offload-arch=gfx950 -O3 -o chan_probe2 chan_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));
constexpr int B = 8192, T = 156, D = 79, NF = 39, NCH = 4;

// cs = candidate stride in floats (T * D for the dense layout)
__global__ __launch_bounds__(768) void k_sweep(float *out, size_t cs, int ntiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) return;
    const int cj = wave - 4;
    const int w = blockIdx.x;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    const int U = ntiles * NCH, per = (U + gridDim.x - 1) / gridDim.x;
    for (int s = 0; s < per; s++) {
        const int u = w * per + s;
        if (u >= U) break;
        const int tile = u / NCH, chunk = (u % NCH + w) % NCH;
        for (int f0 = 0; f0 < NF; f0 += 3)
            for (int half = 0; half < 2; half++) {
                const size_t cand = (size_t)tile * 16 + cj + 8 * half;
                const int f = f0 + fsub;
                if (on && f < NF) {
                    float *p = out + cand * cs + (size_t)(chunk * NF + f) * D + (ql == 19 ? 75 : 4 * ql);
                    f4u v = {1.f, 2.f, 3.f, (float)f};
                    *(f4u *)p = v;
                }
            }
    }
}

static hipEvent_t e0, e1;
static float timeit(std::function<void()> f, int warm = 2, int n = 10) {
    for (int i = 0; i < warm; i++) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < n; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}
static int vmm_alloc(size_t total, size_t chunk, void **out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t n = (total + chunk - 1) / chunk;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
    for (size_t i = 0; i < n; i++) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)va + i * chunk, chunk, 0, h, 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n * chunk, &acc, 1));
    *out = va;
    return 0;
}

int main(int argc, char **argv) {
    const int n_map = argc > 1 ? atoi(argv[1]) : 128;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t NB = (size_t)B * T * D * 4, CS = (size_t)T * D;
    const size_t GB = (size_t)1 << 30;
    auto sweep = [&](float *p, size_t cs = 0) { return timeit([&] { k_sweep<<<256, 768, 150 * 1024>>>(p, cs ? cs : CS, B / 16); }); };
    void *big = nullptr;
    if (vmm_alloc(GB, GB, &big)) { printf("no VMM\n"); CK(hipMalloc(&big, GB)); }
    printf("E1: base offset inside a 1 GiB physical chunk (us)\n  2 MiB steps:");
    for (size_t off = 0; off + NB <= GB; off += (size_t)2 << 20) {
        if ((off >> 21) % 16 == 0) printf("\n   %4zu MiB:", off >> 20);
        printf(" %.1f", sweep((float *)((char *)big + off)));
        fflush(stdout);
    }
    printf("\n  fine steps:");
    for (size_t off : {(size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)262144, (size_t)1048576})
        printf(" +%zu=%.1f", off, sweep((float *)((char *)big + off)));
    printf("\nE6: candidate stride (bytes) on the 1 GiB chunk:");
    for (size_t csb : {(size_t)49296, (size_t)49408, (size_t)49664, (size_t)50176, (size_t)51200, (size_t)53248, (size_t)57344, (size_t)65536, (size_t)66560, (size_t)69632})
        printf(" %zu=%.1f", csb, sweep((float *)big, csb / 4));
    printf("\n");
    fflush(stdout);
    printf("E5: %d exact-size hipMalloc buffers held at once (us):", n_map);
    std::vector<float *> bufs;
    std::vector<float> tus;
    for (int i = 0; i < n_map; i++) {
        float *p = nullptr;
        if (hipMalloc(&p, NB) != hipSuccess) { printf(" [out of memory at %d]", i); break; }
        bufs.push_back(p);
        tus.push_back(sweep(p));
        if (i % 16 == 0) printf("\n   %3d (%p):", i, (void *)p);
        printf(" %.1f", tus.back());
        fflush(stdout);
    }
    printf("\n");
    int fast = -1, slow = -1;
    for (size_t i = 0; i < bufs.size(); i++) { if (tus[i] < 68 && fast < 0) fast = (int)i; if (tus[i] > 76 && slow < 0) slow = (int)i; }
    printf("E2: alternate launches over two buffers\n");
    if (slow >= 0) {
        int slow2 = -1;
        for (size_t i = slow + 1; i < bufs.size(); i++) if (tus[i] > 76) { slow2 = (int)i; break; }
        if (slow2 >= 0) {
            int t = 0;
            const float ta = timeit([&] { k_sweep<<<256, 768, 150 * 1024>>>(bufs[(t++ & 1) ? slow : slow2], CS, B / 16); }, 2, 20);
            printf("  slow %d / slow %d alternating: %.1f us (alone %.1f, %.1f)\n", slow, slow2, ta, tus[slow], tus[slow2]);
        }
    }
    if (fast >= 0) {
        int fast2 = -1;
        for (size_t i = fast + 1; i < bufs.size(); i++) if (tus[i] < 68) { fast2 = (int)i; break; }
        if (fast2 >= 0) {
            int t = 0;
            const float ta = timeit([&] { k_sweep<<<256, 768, 150 * 1024>>>(bufs[(t++ & 1) ? fast : fast2], CS, B / 16); }, 2, 20);
            printf("  fast %d / fast %d alternating: %.1f us (alone %.1f, %.1f)\n", fast, fast2, ta, tus[fast], tus[fast2]);
        }
        if (slow >= 0) {
            int t = 0;
            const float ta = timeit([&] { k_sweep<<<256, 768, 150 * 1024>>>(bufs[(t++ & 1) ? fast : slow], CS, B / 16); }, 2, 20);
            printf("  fast %d / slow %d alternating: %.1f us\n", fast, slow, ta);
        }
        // a fast buffer: is every part of it fast?  quarter-batch runs (2048 candidates = 101 MB) at the four quarters
        printf("  quarters of fast buffer %d:", fast);
        for (int q = 0; q < 4; q++) printf(" %.1f", timeit([&] { k_sweep<<<256, 768, 150 * 1024>>>(bufs[fast] + (size_t)q * 2048 * CS, CS, 128); }));
        printf("\n");
    }
    if (slow >= 0) {
        printf("  quarters of slow buffer %d:", slow);
        for (int q = 0; q < 4; q++) printf(" %.1f", timeit([&] { k_sweep<<<256, 768, 150 * 1024>>>(bufs[slow] + (size_t)q * 2048 * CS, CS, 128); }));
        printf("\n");
    }
    return 0;
}

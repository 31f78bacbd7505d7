// What separates the two placement classes of a large output buffer ("Placement")?
// Hypothesis under test: the size of the page-table FRAGMENTS behind the buffer, i.e. address-translation reach.  The sweep's
// store pattern keeps ~4096 streams 49 KB apart open at once (thousands of pages), a plain fill walks pages one after another.
//  T1  plain hipMalloc buffers held at once: pattern / fill time of each -> one fast and one slow buffer
//  T2  buffers assembled through the virtual-memory API from physical chunks of 64 KiB ... 32 MiB (every chunk mapped by its
//      own hipMemMap, so a translation fragment can be no larger than a chunk): pattern / fill time by chunk size
//  T3  the pattern and the fill on {fast plain, slow plain, smallest-chunk, 2 MiB-chunk} as differently NAMED kernels
//      (k_pattern<0..3>, k_fill<0..3>), ten launches each, so that `rocprofv3 --pmc` rows can be told apart by kernel name.
// Build: hipcc --offload-arch=gfx950 -O3 -o placement_pmc placement_pmc.hip     Run: ./placement_pmc [n_plain=24]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));
constexpr int B = 8192, T = 156, D = 79, NF = 39, NCH = 4;

template <int TAG>
__global__ __launch_bounds__(512) void k_pattern(float *out, int ntiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    const int U = ntiles * NCH, per = (U + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int s = 0; s < per; s++) {
        const int u = blockIdx.x * per + s;
        if (u >= U) break;
        const int tile = u / NCH, chunk = (u % NCH + blockIdx.x) % NCH;
        for (int f0 = 0; f0 < NF; f0 += 3)
            for (int half = 0; half < 2; half++) {
                const size_t cand = (size_t)tile * 16 + wave + 8 * half;
                const int f = f0 + fsub;
                if (on && f < NF) {
                    float *p = out + (cand * T + (size_t)(chunk * NF + f)) * D + (ql == 19 ? 75 : 4 * ql);
                    const f4u v = {0.f, 0.f, 0.f, (float)TAG};
                    *(f4u *)p = v;
                }
            }
    }
}
// the same stores with data that identifies the launch: {tag, quad lane, frame, candidate}
__global__ __launch_bounds__(512) void k_pattern_tag(float *out, int ntiles, float tag) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    const int U = ntiles * NCH, per = (U + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int s = 0; s < per; s++) {
        const int u = blockIdx.x * per + s;
        if (u >= U) break;
        const int tile = u / NCH, chunk = (u % NCH + blockIdx.x) % NCH;
        for (int f0 = 0; f0 < NF; f0 += 3)
            for (int half = 0; half < 2; half++) {
                const size_t cand = (size_t)tile * 16 + wave + 8 * half;
                const int f = f0 + fsub;
                if (on && f < NF) {
                    float *p = out + (cand * T + (size_t)(chunk * NF + f)) * D + (ql == 19 ? 75 : 4 * ql);
                    const f4u v = {tag, (float)ql, (float)(chunk * NF + f), (float)cand};
                    *(f4u *)p = v;
                }
            }
    }
}
// T11: the same stores under other (workgroup, step) -> (tile, chunk) maps; every map writes every (tile, chunk) once
__device__ __forceinline__ int bitrev9(int x) { return (int)(__brev((unsigned)x) >> 23); }
__global__ __launch_bounds__(512) void k_map(float *out, int ntiles, int mode, int prm) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    const int w = blockIdx.x, G = gridDim.x;
    const int nblk = G / NCH, q = (ntiles + nblk - 1) / nblk;   // chunk-stationary maps: nblk workgroups per chunk, q tiles each
    const int U = ntiles * NCH, per = (U + G - 1) / G;
    const int steps = (mode == 0 || mode == 20 || mode == 22 || mode == 30) ? per : mode == 31 ? 9 : q;
    for (int s = 0; s < steps; s++) {
        int tile, chunk;
        if (mode == 0) { const int u = w * per + s; if (u >= U) break; tile = u / NCH; chunk = (u % NCH + w) % NCH; }
        else if (mode == 30) { const int u = w * per + (s + prm * (w % NCH)) % per; if (u >= U) continue; tile = u / NCH; chunk = (u % NCH + w) % NCH; }
        else if (mode == 31) {   // per = 9 on every workgroup: 256 x 9 = 2304 units over 2048: the surplus re-writes units of workgroup w + 1
            const int u = (w * 8 + s) % U; tile = u / NCH; chunk = (u % NCH + w) % NCH; if (s >= prm) break; }
        else if (mode == 20) { const int u = w * per + s; if (u >= U) break; tile = u / NCH; chunk = (u % NCH + prm * w) % NCH; }
        else if (mode == 22) {   // all workgroups on the same chunk at the same time: chunk by chunk over the workgroup's own tiles
            const int tpw = per / NCH; if (tpw * NCH != per) break;
            const int c = s / tpw, t = s % tpw; tile = w * tpw + t; chunk = (c + prm * w) % NCH; if (tile >= ntiles) continue;
        }
        else {
            chunk = (mode == 5) ? w / nblk : w % NCH;
            const int j = (mode == 5) ? w % nblk : w / NCH;
            const int lin = j * q + s;
            if (mode == 1 || mode == 5) tile = lin;
            else if (mode == 2) tile = j * q + (s + j) % q;
            else if (mode == 3) tile = s * nblk + j;
            else if (mode == 4) tile = (int)(((long long)lin * prm) % ntiles);
            else if (mode == 6) tile = (lin + chunk * prm) % ntiles;
            else if (mode == 7) tile = bitrev9(lin) % ntiles;
            else if (mode == 8) tile = (j * q + (s * prm) % q);          // another order inside the block
            else tile = lin;
            if (tile >= ntiles) continue;
        }
        for (int f0 = 0; f0 < NF; f0 += 3)
            for (int half = 0; half < 2; half++) {
                const size_t cand = (size_t)tile * 16 + wave + 8 * half;
                const int f = f0 + fsub;
                if (on && f < NF) {
                    float *p = out + (cand * T + (size_t)(chunk * NF + f)) * D + (ql == 19 ? 75 : 4 * ql);
                    const f4u v = {0.f, 0.f, 0.f, 11.f};
                    *(f4u *)p = v;
                }
            }
    }
}
template <int TAG>
__global__ __launch_bounds__(256) void k_fill(f4 *buf, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const f4 v = {0.f, 0.f, 0.f, (float)TAG}; buf[i] = v; }
}

__global__ __launch_bounds__(256) void k_stamp(f4 *buf, size_t n, float tag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const f4 v = {(float)(i & 1023), 1.f, 2.f, tag}; buf[i] = v; }
}
static hipEvent_t e0, e1;
static float timeit(const std::function<void()> &f, int warm = 2, int n = 8) {
    for (int i = 0; i < warm; i++) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < n; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}
struct vmm_buf { void *va = nullptr; size_t total = 0, chunk = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
static int vmm_alloc(size_t bytes, size_t chunk, vmm_buf *out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t n = (bytes + chunk - 1) / chunk;
    out->total = n * chunk; out->chunk = chunk;
    CK(hipMemAddressReserve(&out->va, out->total, (size_t)2 << 20, nullptr, 0));
    for (size_t i = 0; i < n; i++) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        out->h.push_back(h);
        CK(hipMemMap((char *)out->va + i * chunk, chunk, 0, h, 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(out->va, out->total, &acc, 1));
    return 0;
}
// the same, but the chunks are mapped in a shuffled order (seeded LCG): whatever physical order hipMemCreate handed them out
// in, consecutive parts of the buffer sit in unrelated parts of the card's memory; spread > 1: spread times as many chunks are
// created and every spread-th is used (the others are released after mapping), so that the chunks are not even neighbours
static int vmm_alloc_shuffled(size_t bytes, size_t chunk, int spread, vmm_buf *out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t n = (bytes + chunk - 1) / chunk;
    out->total = n * chunk; out->chunk = chunk;
    CK(hipMemAddressReserve(&out->va, out->total, (size_t)2 << 20, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> all;
    for (size_t i = 0; i < n * spread; i++) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        all.push_back(h);
    }
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; i++) perm[i] = i;
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (size_t i = n; i > 1; i--) { st = st * 6364136223846793005ull + 1442695040888963407ull; std::swap(perm[i - 1], perm[(st >> 33) % i]); }
    for (size_t i = 0; i < n; i++) {
        hipMemGenericAllocationHandle_t h = all[perm[i] * spread];
        out->h.push_back(h);
        CK(hipMemMap((char *)out->va + i * chunk, chunk, 0, h, 0));
    }
    for (size_t i = 0; i < all.size(); i++) if (i % spread) (void)hipMemRelease(all[i]);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(out->va, out->total, &acc, 1));
    return 0;
}
static void vmm_free(vmm_buf *b) {
    if (!b->va) return;
    (void)hipMemUnmap(b->va, b->total);
    for (auto h : b->h) (void)hipMemRelease(h);
    (void)hipMemAddressFree(b->va, b->total);
    b->va = nullptr; b->h.clear();
}

int main(int argc, char **argv) {
    const int n_plain = argc > 1 ? atoi(argv[1]) : 24;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t NBYTES = (size_t)B * T * D * 4, n4 = NBYTES / 16;
    const size_t ALLOC = (NBYTES + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
    const int ntiles = B / 16;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    auto pat = [&](void *p) { return timeit([&] { hipLaunchKernelGGL(k_pattern<9>, dim3(grid), dim3(512), 0, 0, (float *)p, ntiles); }); };
    auto fil = [&](void *p) { return timeit([&] { hipLaunchKernelGGL(k_fill<9>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (f4 *)p, n4); }); };

    // T6 (first, on fresh memory): in-order against shuffled chunk mapping, all buffers of a row held at once
    if (argc > 2 && atoi(argv[2]) != 10) {
        for (size_t c : {(size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
            for (int mode = 0; mode < 3; mode++) {
                std::vector<vmm_buf> held(4);
                printf("T6 chunk %3zu MiB %-22s", c >> 20, mode == 0 ? "in order" : mode == 1 ? "shuffled" : "shuffled, spread 2");
                for (auto &b : held) {
                    if (mode == 0 ? vmm_alloc(NBYTES, c, &b) : vmm_alloc_shuffled(NBYTES, c, mode == 2 ? 2 : 1, &b)) { printf(" alloc failed"); break; }
                    const float f = fil(b.va), q = pat(b.va);
                    printf("  %p %5.1f/%5.1f=%.3f", b.va, q, f, q / f);
                }
                printf("\n");
                for (auto &b : held) vmm_free(&b);
            }
        }
        {   // plain buffers for comparison on the same fresh process
            std::vector<void *> pl;
            printf("T6 plain hipMalloc                  ");
            for (int i = 0; i < 6; i++) { void *p = nullptr; if (hipMalloc(&p, NBYTES) != hipSuccess) break; pl.push_back(p); const float f = fil(p), q = pat(p); printf("  %5.1f/%5.1f=%.3f", q, f, q / f); }
            printf("\n");
            for (void *p : pl) (void)hipFree(p);
        }
        if (atoi(argv[2]) == 2) return 0;
    }
    // T7: the SAME physical chunks mapped at different virtual addresses inside one reserved range
    if (argc > 2 && atoi(argv[2]) == 3) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        const size_t chunk = (size_t)32 << 20, nch = (NBYTES + chunk - 1) / chunk, span = nch * chunk;
        const size_t range = (size_t)24 << 30;
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, range, (size_t)1 << 30, nullptr, 0));
        printf("T7 reserved %zu GiB at %p; %zu chunks of 32 MiB\n", range >> 30, va, nch);
        std::vector<hipMemGenericAllocationHandle_t> hs(nch);
        for (auto &h : hs) CK(hipMemCreate(&h, chunk, &prop, 0));
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        std::vector<size_t> offs;
        if (argc > 3 && atoi(argv[3]) == -1) {   // T9: the same chunks mapped at many virtual addresses AT ONCE (aliases), nothing unmapped in between
            const int NA = 32;
            const size_t stride = (size_t)450 << 20;
            int mapped = 0;
            for (int a = 0; a < NA; a++) {
                char *base = (char *)va + a * stride;
                bool okm = true;
                for (size_t i = 0; i < nch && okm; i++) okm = hipMemMap(base + i * chunk, chunk, 0, hs[i], 0) == hipSuccess;
                if (okm) okm = hipMemSetAccess(base, span, &acc, 1) == hipSuccess;
                if (!okm) { printf("T9 alias %d: mapping failed (%s)\n", a, hipGetErrorString(hipGetLastError())); break; }
                mapped++;
            }
            printf("T9 %d aliases of the same %zu chunks\n", mapped, nch);
            for (int round = 0; round < 3; round++)
                for (int a = 0; a < mapped; a++) {
                    char *base = (char *)va + a * stride;
                    const float tag = 1000.f * round + a + 1;
                    const float f = fil(base);
                    const float q = timeit([&] { hipLaunchKernelGGL(k_pattern_tag, dim3(grid), dim3(512), 0, 0, (float *)base, ntiles, tag); }, 2, 8);
                    float back[4] = {-1.f, -1.f, -1.f, -1.f};
                    CK(hipMemcpy(back, (char *)va + ((size_t)4321 * T + 77) * D * 4, 16, hipMemcpyDeviceToHost));   // read through alias 0
                    printf("T9 round %d alias %2d va %p fill %5.1f pattern %5.1f ratio %.3f  %s\n", round, a, (void *)base, f, q, q / f,
                           back[0] == tag && back[2] == 77.f && back[3] == 4321.f ? "seen through alias 0" : "NOT seen through alias 0");
                }
            return 0;
        }
        if (argc > 4) {   // scan: argv[3] = step in MiB, argv[4] = count
            for (int i = 0; i < atoi(argv[4]); i++) offs.push_back((size_t)i * ((size_t)atoi(argv[3]) << 20));
            for (size_t o : offs) {
                char *base = (char *)va + o;
                for (size_t i = 0; i < nch; i++) CK(hipMemMap(base + i * chunk, chunk, 0, hs[i], 0));
                CK(hipMemSetAccess(base, span, &acc, 1));
                const float f = timeit([&] { hipLaunchKernelGGL(k_fill<9>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (f4 *)base, n4); }, 1, 4);
                const float tag = 1.f + (float)(o >> 20);
                const float q = timeit([&] { hipLaunchKernelGGL(k_pattern_tag, dim3(grid), dim3(512), 0, 0, (float *)base, ntiles, tag); }, 1, 4);
                // did the TIMED stores land?  frame f of candidate c starts with {tag, 0, f, c}
                bool ok = true;
                uint64_t st = 88172645463325252ull + o;
                for (int smp = 0; smp < 24; smp++) {
                    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
                    const size_t cand = smp == 0 ? 0 : smp == 1 ? B - 1 : st % B, fr = smp < 2 ? (smp ? T - 1 : 0) : (st >> 20) % T;
                    float back[4] = {-1.f, -1.f, -1.f, -1.f};
                    CK(hipMemcpy(back, base + (cand * T + fr) * D * 4, 16, hipMemcpyDeviceToHost));
                    const bool okk = back[0] == tag && back[1] == 0.f && back[2] == (float)fr && back[3] == (float)cand;
                    if (!okk && ok) printf("   first bad sample: cand %zu frame %zu: read {%g, %g, %g, %g}, expected {%g, 0, %zu, %zu}\n", cand, fr, back[0], back[1], back[2], back[3], tag, fr, cand);
                    ok = ok && okk;
                }
                printf("T8 %p %5.1f %5.1f %s\n", (void *)base, f, q, ok ? "ok" : "STORES LOST");
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(base, span));
            }
            return 0;
        }
        for (size_t o = 0; o <= ((size_t)16 << 30); o = o ? o * 2 : ((size_t)2 << 20)) offs.push_back(o);
        for (int kq = 1; kq <= 12; kq++) offs.push_back((size_t)kq * ((size_t)388 << 20));
        for (int kq = 1; kq <= 16; kq++) offs.push_back((size_t)kq * ((size_t)2 << 20) + ((size_t)1 << 30));
        offs.push_back(0);
        for (size_t o : offs) {
            char *base = (char *)va + o;
            for (size_t i = 0; i < nch; i++) CK(hipMemMap(base + i * chunk, chunk, 0, hs[i], 0));
            CK(hipMemSetAccess(base, span, &acc, 1));
            const float f = fil(base), q = pat(base);
            printf("T7 offset %8zu MiB (va %p)  fill %5.1f  pattern %5.1f  ratio %.3f\n", o >> 20, (void *)base, f, q, q / f);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(base, span));
        }
        for (auto &h : hs) (void)hipMemRelease(h);
        (void)hipMemAddressFree(va, range);
        return 0;
    }
    // T1
    std::vector<void *> plain; std::vector<float> pr, pp;
    for (int i = 0; i < n_plain; i++) {
        void *p = nullptr;
        if (hipMalloc(&p, ALLOC) != hipSuccess) { (void)hipGetLastError(); break; }
        plain.push_back(p);
        const float f = fil(p), q = pat(p);
        pr.push_back(q / f); pp.push_back(q);
        printf("T1 plain %2d  va %p  fill %6.1f us  pattern %6.1f us  ratio %.3f\n", i, p, f, q, q / f);
    }
    int ifast = 0, islow = 0;
    for (size_t i = 0; i < pr.size(); i++) { if (pr[i] < pr[ifast]) ifast = (int)i; if (pr[i] > pr[islow]) islow = (int)i; }
    printf("T1 fast = plain %d (%.3f), slow = plain %d (%.3f)\n", ifast, pr[ifast], islow, pr[islow]);
    void *fast = plain[ifast], *slow = plain[islow];
    // T10: the same stores from fewer workgroups (and so another unit -> workgroup map): does the slow class need all 256 CUs?
    for (int gr : {256, 252, 248, 240, 232, 228, 224, 208, 192, 171, 160, 128, 103, 64}) {
        const float qf = timeit([&] { hipLaunchKernelGGL(k_pattern<9>, dim3(gr), dim3(512), 0, 0, (float *)fast, ntiles); });
        const float qs = timeit([&] { hipLaunchKernelGGL(k_pattern<9>, dim3(gr), dim3(512), 0, 0, (float *)slow, ntiles); });
        printf("T10 grid %3d (%2d units per workgroup): fast buffer %5.1f us, slow buffer %5.1f us\n", gr, (ntiles * NCH + gr - 1) / gr, qf, qs);
    }
    for (int nt : {512, 513, 514, 520}) {   // one more tile than the buffer's 512 (stays inside the 2 MiB-rounded allocation for 513)
        if ((size_t)nt * 16 * T * D * 4 > ALLOC) continue;
        const float qf = timeit([&] { hipLaunchKernelGGL(k_pattern<9>, dim3(grid), dim3(512), 0, 0, (float *)fast, nt); });
        const float qs = timeit([&] { hipLaunchKernelGGL(k_pattern<9>, dim3(grid), dim3(512), 0, 0, (float *)slow, nt); });
        printf("T10 ntiles %d grid %d: fast buffer %5.1f us, slow buffer %5.1f us\n", nt, grid, qf, qs);
    }
    {
        struct mp { int mode, prm; const char *what; };
        const mp maps[] = {{0, 0, "tile-major: u = w per + s, chunk rotated by w (the probe's pattern)"},
                           {30, 1, "tile-major, the workgroup's units walked from unit w % 4: tile switches staggered"},
                           {30, 3, "tile-major, the workgroup's units walked from unit 3 (w % 4)"},
                           {31, 8, "tile-major, u = 8 w + s, 8 steps (= the first line, other code path)"},
                           {20, 0, "tile-major, chunk = u % 4: every workgroup on the same chunk at a time"},
                           {20, 2, "tile-major, chunk rotated by 2 w: two chunks active at a time"},
                           {22, 0, "chunk by chunk over the workgroup's own tiles, all workgroups on the same chunk"},
                           {22, 2, "chunk by chunk over the workgroup's own tiles, two chunks active at a time"},
                           {22, 1, "chunk by chunk over the workgroup's own tiles, four chunks active at a time"},
                           {1, 0, "chunk-stationary: chunk w % 4, block of q consecutive tiles per workgroup"},
                           {5, 0, "chunk-stationary, chunk = w / 64"},
                           {2, 0, "chunk-stationary, block walked from tile (s + j) % q"},
                           {3, 0, "chunk-stationary, tiles dealt round-robin (s nblk + j)"},
                           {4, 77, "chunk-stationary, tile = 77 lin mod 512"},
                           {4, 3, "chunk-stationary, tile = 3 lin mod 512"},
                           {4, 171, "chunk-stationary, tile = 171 lin mod 512"},
                           {6, 1, "chunk-stationary, blocks shifted by chunk x 1 tile"},
                           {6, 3, "chunk-stationary, blocks shifted by chunk x 3 tiles"},
                           {6, 37, "chunk-stationary, blocks shifted by chunk x 37 tiles"},
                           {7, 0, "chunk-stationary, tile = bit-reversed lin"},
                           {8, 3, "chunk-stationary, block walked in steps of 3"}};
        for (int gr : {256}) {
            for (const mp &m : maps) {
                const float qf = timeit([&] { hipLaunchKernelGGL(k_map, dim3(gr), dim3(512), 0, 0, (float *)fast, ntiles, m.mode, m.prm); });
                const float qs = timeit([&] { hipLaunchKernelGGL(k_map, dim3(gr), dim3(512), 0, 0, (float *)slow, ntiles, m.mode, m.prm); });
                printf("T11 grid %3d  fast %5.1f  slow %5.1f  %s\n", gr, qf, qs, m.what);
            }
        }
    }
    if (argc > 2 && atoi(argv[2]) == 10) return 0;
    for (size_t i = 0; i < plain.size(); i++) if ((int)i != ifast && (int)i != islow) (void)hipFree(plain[i]);

    // T2
    hipMemAllocationProp mp = {};
    mp.type = hipMemAllocationTypePinned; mp.location.type = hipMemLocationTypeDevice; mp.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &mp, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &mp, hipMemAllocationGranularityRecommended));
    printf("T2 allocation granularity: minimum %zu, recommended %zu\n", gmin, grec);
    const size_t chunks[] = {(size_t)64 << 10, (size_t)2 << 20, (size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)512 << 20};
    vmm_buf small, two;
    for (size_t c : chunks) {
        if (c < gmin) { printf("T2 chunk %zu KiB below the minimum granularity, skipped\n", c >> 10); continue; }
        for (int rep = 0; rep < 4; rep++) {
            vmm_buf b;
            if (vmm_alloc(NBYTES, c, &b)) { printf("T2 chunk %zu KiB: allocation failed\n", c >> 10); break; }
            const float f = fil(b.va), q = pat(b.va);
            printf("T2 chunk %6zu KiB (%5zu maps)  fill %6.1f us  pattern %6.1f us  ratio %.3f\n", c >> 10, b.h.size(), f, q, q / f);
            if (!small.va && rep == 0) small = b;
            else if (c == ((size_t)2 << 20) && !two.va) two = b;
            else vmm_free(&b);
        }
    }

    // T3
    void *bufs[4] = {fast, slow, small.va ? small.va : slow, two.va ? two.va : fast};
    const char *names[4] = {"fast plain", "slow plain", "smallest chunks", "2 MiB chunks"};
    for (int rep = 0; rep < 10; rep++) {
        hipLaunchKernelGGL(k_pattern<0>, dim3(grid), dim3(512), 0, 0, (float *)bufs[0], ntiles);
        hipLaunchKernelGGL(k_pattern<1>, dim3(grid), dim3(512), 0, 0, (float *)bufs[1], ntiles);
        hipLaunchKernelGGL(k_pattern<2>, dim3(grid), dim3(512), 0, 0, (float *)bufs[2], ntiles);
        hipLaunchKernelGGL(k_pattern<3>, dim3(grid), dim3(512), 0, 0, (float *)bufs[3], ntiles);
        hipLaunchKernelGGL(k_fill<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (f4 *)bufs[0], n4);
        hipLaunchKernelGGL(k_fill<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (f4 *)bufs[1], n4);
        hipLaunchKernelGGL(k_fill<2>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (f4 *)bufs[2], n4);
        hipLaunchKernelGGL(k_fill<3>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (f4 *)bufs[3], n4);
    }
    CK(hipDeviceSynchronize());
    for (int t = 0; t < 4; t++) printf("T3 tag %d = %s: pattern %6.1f us, fill %6.1f us\n", t, names[t], pat(bufs[t]), fil(bufs[t]));
    // T4: does the class survive an idle gap between launches (caches written back, nothing in flight)?  per-launch events
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int t = 0; t < 2; t++) {
            for (int gap_us : {0, 200, 2000}) {
                float acc = 0.f;
                for (int i = 0; i < 12; i++) {
                    if (gap_us) { CK(hipDeviceSynchronize()); const auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < gap_us) {} }
                    CK(hipEventRecord(a));
                    hipLaunchKernelGGL(k_pattern<9>, dim3(grid), dim3(512), 0, 0, (float *)bufs[t], ntiles);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms = 0.f; CK(hipEventElapsedTime(&ms, a, b));
                    if (i >= 2) acc += ms;
                }
                printf("T4 %s, idle gap %4d us before every launch: pattern %6.1f us per launch (event pair)\n", names[t], gap_us, acc * 1e3f / 10);
            }
        }
    }
    vmm_free(&small); vmm_free(&two);
    (void)hipFree(fast); if (slow != fast) (void)hipFree(slow);
    // T5: ONE large allocation, the pattern on 385 MiB windows at 256 MiB steps: where do the classes change?
    {
        size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
        const size_t big = std::min<size_t>((size_t)32 << 30, fr / 2);
        void *p = nullptr;
        if (hipMalloc(&p, big) == hipSuccess) {
            printf("T5 one allocation of %zu MiB at %p\n", big >> 20, p);
            for (size_t off = 0; off + NBYTES <= big; off += (size_t)256 << 20) {
                const float f = fil((char *)p + off), q = pat((char *)p + off);
                printf("T5 offset %6zu MiB  fill %6.1f  pattern %6.1f  ratio %.3f %s\n", off >> 20, f, q, q / f, q / f > 1.15f ? "slow" : "");
            }
            (void)hipFree(p);
        } else { (void)hipGetLastError(); printf("T5 allocation failed\n"); }
    }
    printf("done\n");
    return 0;
}

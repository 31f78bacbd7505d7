// Two questions about WHERE the frames kernel's output lives ("Placement"):
//  1. which address bits select the memory channel?  A fill restricted to the granules whose index has even parity
//     under a bit mask M uses half of the channels if M is (part of) a channel-select group: its rate halves.
//  2. does any unit -> workgroup schedule of the sweep's store pattern run at the fill rate on PHYSICALLY CONTIGUOUS
//     memory (one virtual-memory chunk), where the present schedule is always in the slow class?
// Build: hipcc --offload-arch=gfx950 -O3 -o chan_probe chan_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));

// ---- 1. parity-restricted fill: granule = 128 bytes (8 lanes x 16 B) -----------------------------------------
__global__ __launch_bounds__(256) void k_mask(f4 *buf, unsigned n_half, unsigned M, int p) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned j = idx >> 3, sub = idx & 7;
    if (j >= n_half) return;
    unsigned i;
    if (M == 0) i = j;
    else {
        const unsigned lo = j & ((1u << p) - 1), hi = j >> p;
        i = (hi << (p + 1)) | lo;
        i |= (unsigned)(__popc(i & M) & 1) << p;
    }
    f4 v = {1.f, 2.f, 3.f, (float)i};
    buf[(size_t)i * 8 + sub] = v;
}

// ---- 2. the sweep's store pattern under a schedule -------------------------------------------------------------
constexpr int B = 8192, T = 156, D = 79, NF = 39, NCH = 4, NTILES = B / 16;
struct sched {
    int mode;       // 0: workgroup w owns units 8w..8w+7 (tiles 2w, 2w+1), 1: tiles w and w + 256, 2: step s = units 256 s + w
    int A;          // tile' = tile * A mod 512 (A odd)
    int skew_m, skew_cyc;   // workgroup w starts (w % skew_m) * skew_cyc cycles late
    int wavemap;    // 0: wave holds candidates cj, cj + 8; 1: 2 cj, 2 cj + 1
    int rot;        // 1: chunk rotation by w % 4 (as the kernel does)
};
__global__ __launch_bounds__(768) void k_sweep(float *out, sched sc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) return;
    const int cj = wave - 4;
    const int w = blockIdx.x;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    if (sc.skew_m > 1) {
        const long long t0 = __builtin_amdgcn_s_memtime();
        const long long wait = (long long)(w % sc.skew_m) * sc.skew_cyc;
        while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    for (int s = 0; s < 8; s++) {
        int tile, chunk;
        if (sc.mode == 0) { tile = 2 * w + s / 4; chunk = s % 4; }
        else if (sc.mode == 1) { tile = w + 256 * (s / 4); chunk = s % 4; }
        else if (sc.mode == 2) { const int u = 256 * s + w; tile = u / 4; chunk = u % 4; }
        else if (sc.mode == 3) { chunk = w % 4; tile = 8 * (w / 4) + s; }        // chunk-stationary, a block of 8 tiles per group of 4 workgroups
        else if (sc.mode == 4) { chunk = w % 4; tile = (w / 4) + 64 * s; }
        else if (sc.mode == 5) { chunk = w / 64; tile = 8 * (w % 64) + s; }
        else if (sc.mode == 6) { chunk = w / 64; tile = (w % 64) + 64 * s; }
        else { chunk = w % 4; tile = 8 * (w / 4) + ((s + w / 4) & 7); }          // mode 3 with the block walked from a group-specific start
        if (sc.rot) chunk = (chunk + w) % 4;
        tile = (int)(((long long)tile * sc.A) % NTILES);
        for (int f0 = 0; f0 < NF; f0 += 3)
            for (int half = 0; half < 2; half++) {
                const int k16 = sc.wavemap ? 2 * cj + half : cj + 8 * half;
                const size_t cand = (size_t)tile * 16 + k16;
                const int f = f0 + fsub;
                if (on && f < NF) {
                    float *p = out + (cand * T + chunk * NF + f) * D + (ql == 19 ? 75 : 4 * ql);
                    f4u v = {1.f, 2.f, 3.f, (float)f};
                    *(f4u *)p = v;
                }
            }
    }
}
__global__ __launch_bounds__(256) void k_fill(f4 *buf, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { f4 v = {1.f, 2.f, 3.f, 4.f}; buf[i] = v; }
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
    const int do_mask = argc > 1 ? atoi(argv[1]) : 1, do_sched = argc > 2 ? atoi(argv[2]) : 1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t GB = (size_t)1 << 30;
    void *big = nullptr;
    if (vmm_alloc(GB, GB, &big)) { printf("no VMM: falling back to hipMalloc\n"); CK(hipMalloc(&big, GB)); }
    printf("1 GiB buffer (one physical chunk) at %p\n", big);
    if (do_mask) {
        const int NBITS = 23;   // granule index bits 0..22 = address bits 7..29
        const unsigned n_half = 1u << (NBITS - 1);
        auto run = [&](unsigned M) {
            int p = 0;
            while (M && !((M >> p) & 1)) p++;
            const unsigned threads = n_half * 8;
            return timeit([&] { k_mask<<<threads / 256, 256>>>((f4 *)big, n_half, M, p); }, 1, 4);
        };
        const double bytes = (double)n_half * 128;
        printf("parity-restricted fill of %.0f MB (us; GB/s):\n", bytes / 1e6);
        { const float t = run(0); printf("  lower half, unrestricted: %.1f us  %.0f GB/s\n", t, bytes / t / 1e3); }
        std::vector<float> single(NBITS);
        for (int b = 0; b < NBITS; b++) {
            single[b] = run(1u << b);
            printf("  addr bit %2d fixed: %7.1f us  %6.0f GB/s\n", b + 7, single[b], bytes / single[b] / 1e3);
        }
        printf("pairs (addr bits i ^ j = 0), GB/s; rows i, columns j > i:\n      ");
        for (int j = 0; j < NBITS; j++) printf(" %4d", j + 7);
        printf("\n");
        for (int i = 0; i < NBITS; i++) {
            printf("  %2d: ", i + 7);
            for (int j = 0; j < NBITS; j++) {
                if (j <= i) { printf("    ."); continue; }
                const float t = run((1u << i) | (1u << j));
                printf(" %4.0f", bytes / t / 1e4);   // in units of 10 GB/s
            }
            printf("\n"); fflush(stdout);
        }
    }
    if (do_sched) {
        const size_t NB = (size_t)B * T * D * 4;
        std::vector<float *> bufs;
        std::vector<const char *> names;
        bufs.push_back((float *)big); names.push_back("vmm-1GiB-chunk");
        for (int i = 0; i < 12; i++) { float *p; CK(hipMalloc(&p, NB + 4096)); bufs.push_back(p); names.push_back("hipMalloc"); }
        { void *p = nullptr; if (!vmm_alloc(NB, (size_t)2 << 20, &p)) { bufs.push_back((float *)p); names.push_back("vmm-2MiB-chunks"); } }
        CK(hipFuncSetAttribute((const void *)k_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (size_t bi = 0; bi < bufs.size(); bi++) {
            float *buf = bufs[bi];
            const float tf = timeit([&] { k_fill<<<(unsigned)((NB / 16 + 255) / 256), 256>>>((f4 *)buf, NB / 16); });
            printf("== buffer %zu (%s): fill %.1f us\n", bi, names[bi], tf);
            for (int mode = 0; mode < 8; mode++)
                for (int rot = 0; rot < 1; rot++)
                    for (int wm = 0; wm < 1; wm++) {
                        printf("  mode %d rot %d wavemap %d | A:", mode, rot, wm);
                        for (int A : {1, 1, 3, 37}) {
                            sched sc = {mode, A, 1, 0, wm, rot};
                            printf(" %d=%.1f", A, timeit([&] { k_sweep<<<256, 768, 150 * 1024>>>(buf, sc); }));
                        }
                        printf("\n"); fflush(stdout);
                    }
            for (int m : {0}) {
                printf("  skew m=%3d | cycles:", m);
                for (int cyc : {500, 2000, 8000, 20000}) {
                    if ((long long)m * cyc > 400000) continue;
                    sched sc = {0, 1, m, cyc, 0, 1};
                    printf(" %d=%.1f", cyc, timeit([&] { k_sweep<<<256, 768, 150 * 1024>>>(buf, sc); }));
                }
                printf("\n"); fflush(stdout);
            }
        }
    }
    return 0;
}

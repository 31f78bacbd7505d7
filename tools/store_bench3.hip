// Does the ORDER in which a persistent grid walks its (candidate tile, time chunk) units change the store bandwidth?
// The frames kernel's sweep pattern (8 waves per workgroup, wave = 2 candidates, 948-byte unaligned quad-row pieces,
// one 768-thread workgroup per CU), stores only, under different unit -> workgroup mappings.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bench3 store_bench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int B = 8192, T = 156, D = 79, NT = 28, NCH = 6, NTILES = B / 16;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef f3 f3u __attribute__((aligned(4)));

// mode 0: blocked (workgroup w owns units [w U/G, (w+1) U/G), tile-major)        -- the kernel today
// mode 1: interleaved units (step s: unit s G + w)                                -- a compact moving window of G units
// mode 2: interleaved tiles (workgroup w owns tiles w, w + G, ...; 6 chunks each)
// mode 3: chunk-major (all tiles' chunk 0 first: unit order chunk * NTILES + tile, blocked)
// The same walk (blocked units), but each wave writes its candidate's chunk as a FLAT run of the pattern `pat`:
//  1: 16-byte aligned float4 per lane, 1 KB per wave instruction     2: dword per lane, 256 B per instruction
//  3: aligned float4, 60 lanes (960 B)                               4: float2 per lane (8-byte aligned), 512 B
//  5: unaligned float4 flat (start shifted by 4 bytes), 1 KB
__global__ __launch_bounds__(768) void k_flat_sweep(float *out, int pat, int lds_touch) {
    extern __shared__ float dyn[];
    if (lds_touch < 0) dyn[threadIdx.x] = 1.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) return;
    const int cj = wave - 4;
    const int G = gridDim.x, U = NTILES * NCH, w = blockIdx.x;
    const int per = (U + G - 1) / G;
    for (int s = 0; s < per; s++) {
        const int u = w * per + s;
        if (u >= U) continue;
        const int tile = u / NCH, chunk = u % NCH;
        for (int half = 0; half < 2; half++) {
            const size_t e0 = ((size_t)(tile * 16 + cj + 8 * half) * T + chunk * 26) * D;   // first float of the piece
            const size_t e1 = e0 + (size_t)26 * D;
            if (pat == 1 || pat == 3 || pat == 5) {
                const int nl = pat == 3 ? 60 : 64;
                size_t a = (e0 + 3) / 4 * 4 + (pat == 5 ? 1 : 0);
                for (size_t i = a + 4 * lane; i + 4 <= e1; i += 4 * nl)
                    if (lane < nl) { f4u v = {1.f, 2.f, 3.f, 4.f}; *(f4u *)(out + i) = v; }
            } else if (pat == 2) {
                for (size_t i = e0 + lane; i < e1; i += 64) out[i] = 1.5f;
            } else {
                size_t a = (e0 + 1) / 2 * 2;
                for (size_t i = a + 2 * lane; i + 2 <= e1; i += 128) *(float2 *)(out + i) = make_float2(1.f, 2.f);
            }
        }
    }
}

// All 8 sweep waves of a workgroup write ONE candidate's chunk together: the unit's 16 x ceil(26/3) row groups
// (3 rows = 948 bytes each) are dealt round-robin to the waves, so at any moment the workgroup writes ~7.6 KB of
// consecutive addresses instead of 16 separate streams.  coop = 1: groups dealt wave by wave; coop = 2: two
// consecutive groups per wave (1896 bytes per wave visit).
__global__ __launch_bounds__(768) void k_coop_sweep(float *out, int coop, int lds_touch) {
    extern __shared__ float dyn[];
    if (lds_touch < 0) dyn[threadIdx.x] = 1.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) return;
    const int cj = wave - 4;
    const int G = gridDim.x, U = NTILES * NCH, w = blockIdx.x;
    const int per = (U + G - 1) / G;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    const int d0 = 4 * ql, nst = ql == 19 ? 3 : 4;
    const int nT = 26, ngr = (nT + 2) / 3;           // 9 row groups per candidate
    for (int s = 0; s < per; s++) {
        const int u = w * per + s;
        if (u >= U) continue;
        const int tile = u / NCH, chunk = u % NCH;
        const int total = 16 * ngr;
        for (int g0 = cj * coop; g0 < total; g0 += 8 * coop)
            for (int k = 0; k < coop; k++) {
                const int gidx = g0 + k;
                if (gidx >= total) break;
                const int cand = gidx / ngr, grp = gidx - cand * ngr;
                const int f = grp * 3 + fsub;
                float *base = out + ((size_t)(tile * 16 + cand) * T + chunk * 26) * D;
                if (on && f < nT) {
                    float *p = base + (size_t)f * D + d0;
                    if (nst == 4) { f4u v = {1.f, 2.f, 3.f, (float)f}; *(f4u *)p = v; }
                    else { f3u v = {1.f, 2.f, 3.f}; *(f3u *)p = v; }
                }
            }
    }
}

__global__ __launch_bounds__(768) void k_sweep(float *out, int mode, int lds_touch) {
    extern __shared__ float dyn[];
    if (lds_touch < 0) dyn[threadIdx.x] = 1.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) return;                       // the producer waves store nothing
    const int cj = wave - 4;
    const int G = gridDim.x, U = NTILES * NCH, w = blockIdx.x;
    const int per = (U + G - 1) / G;
    const int fsub = lane / 20, ql = lane % 20;  // 3 rows x 20 quads (the last quad of a row: 3 floats)
    const bool on = lane < 60;
    const int d0 = 4 * ql, nst = ql == 19 ? 3 : 4;
    for (int s = 0; s < per; s++) {
        int u;
        if (mode == 0 || mode == 3) u = w * per + s;
        else if (mode == 1) u = s * G + w;
        else { const int tl = (s / NCH) * G + w; u = tl * NCH + s % NCH; if (tl >= NTILES) continue; }
        if (u >= U) continue;
        int tile, chunk;
        if (mode == 3) { chunk = u / NTILES; tile = u % NTILES; } else { tile = u / NCH; chunk = u % NCH; }
        const int t0 = chunk * NT - (chunk > 0 ? 2 * chunk : 0);   // 6 chunks of 26 frames cover 156
        const int nT = 26;
        for (int half = 0; half < 2; half++) {
            float *base = out + ((size_t)(tile * 16 + cj + 8 * half) * T + (t0 < 0 ? 0 : (chunk * 26))) * D;
            for (int f0 = 0; f0 < nT; f0 += 3) {
                const int f = f0 + fsub;
                if (on && f < nT) {
                    float *p = base + (size_t)f * D + d0;
                    if (nst == 4) { f4u v = {1.f, 2.f, 3.f, (float)f}; *(f4u *)p = v; }
                    else { f3u v = {1.f, 2.f, 3.f}; *(f3u *)p = v; }
                }
            }
        }
    }
}

int main() {
    float *out;
    const size_t N = (size_t)B * T * D;
    CK(hipMalloc(&out, N * 4));
    CK(hipFuncSetAttribute((const void *)k_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_coop_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_flat_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](std::function<void()> f) {
        for (int i = 0; i < 3; i++) f();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms / 20;
    };
    const char *pn[] = {"", "flat aligned float4 (1 KB / instr)", "flat dword (256 B / instr)", "flat aligned float4, 60 lanes", "flat float2 (512 B / instr)", "flat UNALIGNED float4"};
    for (int rep = 0; rep < 2; rep++)
        for (int pat = 1; pat <= 5; pat++) {
            float ms = timeit([&] { k_flat_sweep<<<256, 768, 150 * 1024>>>(out, pat, -1); });
            printf("%-36s lds 150K  %7.1f us  %7.1f GB/s\n", pn[pat], ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
            fflush(stdout);
        }
    for (int rep = 0; rep < 2; rep++)
        for (int coop = 1; coop <= 2; coop++) {
            float ms = timeit([&] { k_coop_sweep<<<256, 768, 150 * 1024>>>(out, coop, -1); });
            printf("cooperative sweep, %d group(s) per visit  lds 150K  %7.1f us  %7.1f GB/s\n", coop, ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
            fflush(stdout);
        }
    const char *names[] = {"blocked (today)", "interleaved units (moving window)", "interleaved tiles", "chunk-major blocked"};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 4; mode++)
            for (int lds : {0, 150 * 1024}) {
                float ms = timeit([&] { k_sweep<<<256, 768, lds>>>(out, mode, lds ? -1 : 0); });
                printf("%-36s lds %3dK  %7.1f us  %7.1f GB/s\n", names[mode], lds / 1024, ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
                fflush(stdout);
            }
    return 0;
}

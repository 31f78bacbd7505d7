// Store-path microbenchmark (MI355X): how fast can 8192 x 156 x 79 float32 frames be written
// under different thread->address mappings?  Build: hipcc --offload-arch=gfx950 -O3 -o store_bench store_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int B = 8192, T = 156, D = 79, TD = T * D;

// A: float4 per lane, fully contiguous grid-stride
__global__ void k_f4(float4 *out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// B: dword per lane, fully contiguous grid-stride
__global__ void k_dw(float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)i;
}
// C: block = (16 candidates, chunk of NT frames); thread = flat element of the chunk, loops candidates
__global__ void k_flat(float *out, int NT) {
    int nch = (T + NT - 1) / NT;
    int tile = blockIdx.x / nch, ch = blockIdx.x % nch;
    int t0 = ch * NT, nT = min(NT, T - t0);
    float *base = out + (size_t)tile * 16 * TD + (size_t)t0 * D;
    for (int o = threadIdx.x; o < nT * D; o += blockDim.x)
        for (int c = 0; c < 16; c++) base[(size_t)c * TD + o] = (float)o;
}
// D: block = (16 candidates, chunk); thread = (cand, d) pair, walks frames (the v2 kernel's mapping)
__global__ void k_pair(float *out, int NT) {
    extern __shared__ float dyn[];
    if (NT < 0) dyn[threadIdx.x] = 1.f;
    int nch = (T + NT - 1) / NT;
    int tile = blockIdx.x / nch, ch = blockIdx.x % nch;
    int t0 = ch * NT, nT = min(NT, T - t0);
    float *base = out + (size_t)tile * 16 * TD + (size_t)t0 * D;
    for (int p = threadIdx.x; p < 16 * D; p += blockDim.x) {
        int c = p / D, d = p - c * D;
        float *q = base + (size_t)c * TD + d;
        for (int f = 0; f < nT; f++) q[(size_t)f * D] = (float)f;
    }
}
// E: like D but a wave never straddles candidates: wave w of the block owns candidates w, w+4, ...; lanes = d (two passes over d)
__global__ void k_pair_aligned(float *out, int NT) {
    int nch = (T + NT - 1) / NT;
    int tile = blockIdx.x / nch, ch = blockIdx.x % nch;
    int t0 = ch * NT, nT = min(NT, T - t0);
    float *base = out + (size_t)tile * 16 * TD + (size_t)t0 * D;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = wave; c < 16; c += 4)
        for (int d = lane; d < D; d += 64) {
            float *q = base + (size_t)c * TD + d;
            for (int f = 0; f < nT; f++) q[(size_t)f * D] = (float)f;
        }
}
// F: block = (1 candidate x all frames) contiguous 49 KB, dword per lane
__global__ void k_cand(float *out) {
    float *base = out + (size_t)blockIdx.x * TD;
    for (int o = threadIdx.x; o < TD; o += blockDim.x) base[o] = (float)o;
}
// G: block = (16 candidates, chunk); each wave owns 4 candidates; flat over the chunk (contiguous 256 B per store, one candidate at a time)
__global__ void k_wavecand(float *out, int NT) {
    extern __shared__ float dyn[];
    if (NT < 0) dyn[threadIdx.x] = 1.f;
    int nch = (T + NT - 1) / NT;
    int tile = blockIdx.x / nch, ch = blockIdx.x % nch;
    int t0 = ch * NT, nT = min(NT, T - t0);
    float *base = out + (size_t)tile * 16 * TD + (size_t)t0 * D;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = wave; c < 16; c += 4)
        for (int o = lane; o < nT * D; o += 64) base[(size_t)c * TD + o] = (float)o;
}

// H: quad-row mapping: wave = one candidate at a time; lane = (row in group of 3, quad of 4 channels);
//    unaligned dwordx4 stores, last quad of a row is a dwordx3 store; 948 contiguous bytes per wave store
__global__ void k_quadrow(float *out, int NT) {
    extern __shared__ float dyn[];
    if (NT < 0) dyn[threadIdx.x] = 1.f;
    int nch = (T + NT - 1) / NT;
    int tile = blockIdx.x / nch, ch = blockIdx.x % nch;
    int t0 = ch * NT, nT = min(NT, T - t0);
    float *base = out + (size_t)tile * 16 * TD + (size_t)t0 * D;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int fsub = lane / 20, q = lane - fsub * 20;
    for (int c = wave; c < 16; c += 4) {
        float *cb = base + (size_t)c * TD;
        for (int f0 = 0; f0 < nT; f0 += 3) {
            int f = f0 + fsub;
            if (lane < 60 && f < nT) {
                float *p = cb + (size_t)f * D + 4 * q;
                float v = (float)f;
                if (q < 19) {
                    typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
                    *(f4 *)p = (f4){v, v, v, v};
                } else {
                    p[0] = v; p[1] = v; p[2] = v;
                }
            }
        }
    }
}
// I: quad-row with the first quad storing only its 4th element (root channels written elsewhere)
__global__ void k_quadrow_noroot(float *out, int NT) {
    int nch = (T + NT - 1) / NT;
    int tile = blockIdx.x / nch, ch = blockIdx.x % nch;
    int t0 = ch * NT, nT = min(NT, T - t0);
    float *base = out + (size_t)tile * 16 * TD + (size_t)t0 * D;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int fsub = lane / 20, q = lane - fsub * 20;
    for (int c = wave; c < 16; c += 4) {
        float *cb = base + (size_t)c * TD;
        for (int f0 = 0; f0 < nT; f0 += 3) {
            int f = f0 + fsub;
            if (lane < 60 && f < nT) {
                float *p = cb + (size_t)f * D + 4 * q;
                float v = (float)f;
                if (q == 0) p[3] = v;
                else if (q < 19) {
                    typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
                    *(f4 *)p = (f4){v, v, v, v};
                } else {
                    p[0] = v; p[1] = v; p[2] = v;
                }
            }
        }
        // root channels: 3 x nT scattered dwords per candidate
        for (int i = lane; i < 3 * nT; i += 64) cb[(size_t)(i / 3) * D + (i % 3)] = 1.0f;
    }
}

template <typename F>
float timeit(F launch, int iters = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    hipFuncSetAttribute((const void*)k_pair, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024);
    hipFuncSetAttribute((const void*)k_wavecand, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024);
    hipFuncSetAttribute((const void*)k_quadrow, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024);
    size_t n = (size_t)B * TD;
    float *out; CK(hipMalloc(&out, n * 4));
    double gb = n * 4 / 1e9;
    auto rep = [&](const char *name, float ms) { printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, gb / (ms * 1e-3)); };
    rep("A float4 contiguous (2048 blk)", timeit([&] { k_f4<<<2048, 256>>>((float4 *)out, n / 4); }));
    rep("A float4 contiguous (8192 blk)", timeit([&] { k_f4<<<8192, 256>>>((float4 *)out, n / 4); }));
    rep("B dword contiguous (2048 blk)", timeit([&] { k_dw<<<2048, 256>>>(out, n); }));
    rep("B dword contiguous (16384 blk)", timeit([&] { k_dw<<<16384, 256>>>(out, n); }));
    for (int NT : {28}) {
        int nch = (T + NT - 1) / NT, grid = B / 16 * nch;
        char nm[64];
        snprintf(nm, 64, "C flat x16cand NT=%d", NT); rep(nm, timeit([&] { k_flat<<<grid, 256>>>(out, NT); }));
        snprintf(nm, 64, "D pair-walk NT=%d", NT); rep(nm, timeit([&] { k_pair<<<grid, 256>>>(out, NT); }));
        snprintf(nm, 64, "E pair-walk wave=cand NT=%d", NT); rep(nm, timeit([&] { k_pair_aligned<<<grid, 256>>>(out, NT); }));
        snprintf(nm, 64, "G flat wave=cand NT=%d", NT); rep(nm, timeit([&] { k_wavecand<<<grid, 256>>>(out, NT); }));
        snprintf(nm, 64, "H quad-row x4 unaligned NT=%d", NT); rep(nm, timeit([&] { k_quadrow<<<grid, 256>>>(out, NT); }));
        for (int lds : {20 * 1024, 40 * 1024, 53 * 1024, 80 * 1024}) {
            snprintf(nm, 64, "  D lds=%dK", lds / 1024); rep(nm, timeit([&] { k_pair<<<grid, 256, lds>>>(out, NT); }));
            snprintf(nm, 64, "  G lds=%dK", lds / 1024); rep(nm, timeit([&] { k_wavecand<<<grid, 256, lds>>>(out, NT); }));
            snprintf(nm, 64, "  H lds=%dK", lds / 1024); rep(nm, timeit([&] { k_quadrow<<<grid, 256, lds>>>(out, NT); }));
        }
        snprintf(nm, 64, "I quad-row + root scatter NT=%d", NT); rep(nm, timeit([&] { k_quadrow_noroot<<<grid, 256>>>(out, NT); }));
    }
    rep("F one candidate per block", timeit([&] { k_cand<<<B, 256>>>(out); }));
    hipFree(out);
    return 0;
}

// Why does torch's fill_ write 404 MB in 61 us when grid-stride store loops need 70-96 us?  Write-only patterns over
// the same 8192 x 156 x 79 float32 buffer.  Build: hipcc --offload-arch=gfx950 -O3 -o store_bench2 store_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr size_t N = (size_t)8192 * 156 * 79;

// one float4 per thread, no loop (torch-like when V = 1); V float4 per thread, block-contiguous per pass
template <int V>
__global__ void k_oneshot_f4(float4 *out, size_t n4) {
    size_t base = (size_t)blockIdx.x * blockDim.x * V + threadIdx.x;
#pragma unroll
    for (int v = 0; v < V; v++) {
        size_t i = base + (size_t)v * blockDim.x;
        if (i < n4) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
template <int V>
__global__ void k_oneshot_dw(float *out, size_t n) {
    size_t base = (size_t)blockIdx.x * blockDim.x * V + threadIdx.x;
#pragma unroll
    for (int v = 0; v < V; v++) {
        size_t i = base + (size_t)v * blockDim.x;
        if (i < n) out[i] = 1.5f;
    }
}
// persistent: each block owns a contiguous slab and walks it (the shape of the frames kernel's sweep)
__global__ void k_slab_dw(float *out, size_t n) {
    size_t per = (n + gridDim.x - 1) / gridDim.x;
    size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (size_t i = b + threadIdx.x; i < e; i += blockDim.x) out[i] = 1.5f;
}
__global__ void k_slab_f4(float4 *out, size_t n4) {
    size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    size_t b = (size_t)blockIdx.x * per, e = b + per < n4 ? b + per : n4;
    for (size_t i = b + threadIdx.x; i < e; i += blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// persistent, each WAVE owns a contiguous slab (8 waves of a 512-thread block = 8 streams per CU)
__global__ void k_waveslab_dw(float *out, size_t n) {
    size_t nw = (size_t)gridDim.x * (blockDim.x >> 6);
    size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    size_t per = (n + nw - 1) / nw;
    size_t b = w * per, e = b + per < n ? b + per : n;
    for (size_t i = b + (threadIdx.x & 63); i < e; i += 64) out[i] = 1.5f;
}
// grid-stride
__global__ void k_stride_dw(float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.5f;
}
__global__ void k_stride_f4(float4 *out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    float *out;
    CK(hipMalloc(&out, N * 4));
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
    auto rep = [&](const char *name, float ms) { printf("%-44s %7.1f us  %7.1f GB/s\n", name, ms * 1e3, N * 4 / 1e9 / (ms * 1e-3)); fflush(stdout); };
    const size_t n4 = N / 4;
    char nm[96];
    rep("memset (hipMemsetAsync)", timeit([&] { hipMemsetAsync(out, 0, N * 4, 0); }));
    rep("one float4 per thread, block 256", timeit([&] { k_oneshot_f4<1><<<(unsigned)((n4 + 255) / 256), 256>>>((float4 *)out, n4); }));
    rep("4 float4 per thread, block 256", timeit([&] { k_oneshot_f4<4><<<(unsigned)((n4 + 1023) / 1024), 256>>>((float4 *)out, n4); }));
    rep("16 float4 per thread, block 256", timeit([&] { k_oneshot_f4<16><<<(unsigned)((n4 + 4095) / 4096), 256>>>((float4 *)out, n4); }));
    rep("one dword per thread, block 256", timeit([&] { k_oneshot_dw<1><<<(unsigned)((N + 255) / 256), 256>>>(out, N); }));
    rep("4 dword per thread, block 256", timeit([&] { k_oneshot_dw<4><<<(unsigned)((N + 1023) / 1024), 256>>>(out, N); }));
    rep("16 dword per thread, block 256", timeit([&] { k_oneshot_dw<16><<<(unsigned)((N + 4095) / 4096), 256>>>(out, N); }));
    rep("64 dword per thread, block 256", timeit([&] { k_oneshot_dw<64><<<(unsigned)((N + 16383) / 16384), 256>>>(out, N); }));
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        snprintf(nm, 96, "grid-stride dword, %d x 256", blocks); rep(nm, timeit([&] { k_stride_dw<<<blocks, 256>>>(out, N); }));
    }
    for (int blocks : {256, 1024, 2048}) {
        snprintf(nm, 96, "grid-stride float4, %d x 256", blocks); rep(nm, timeit([&] { k_stride_f4<<<blocks, 256>>>((float4 *)out, n4); }));
    }
    for (int threads : {256, 512, 1024}) {
        snprintf(nm, 96, "block slab dword, 256 x %d", threads); rep(nm, timeit([&] { k_slab_dw<<<256, threads>>>(out, N); }));
        snprintf(nm, 96, "block slab float4, 256 x %d", threads); rep(nm, timeit([&] { k_slab_f4<<<256, threads>>>((float4 *)out, n4); }));
    }
    for (int blocks : {256, 512, 1024}) {
        snprintf(nm, 96, "block slab dword, %d x 512", blocks); rep(nm, timeit([&] { k_slab_dw<<<blocks, 512>>>(out, N); }));
    }
    rep("wave slab dword, 256 x 512 (8 streams/CU)", timeit([&] { k_waveslab_dw<<<256, 512>>>(out, N); }));
    rep("wave slab dword, 256 x 1024 (16 streams/CU)", timeit([&] { k_waveslab_dw<<<256, 1024>>>(out, N); }));
    return 0;
}

// Is the "fast region" of device memory (tools/placement_probe3.py) a property of the memory or of the frames kernel?
// Eight separate allocations of the output's size, each written by (a) a grid-stride fill and (b) per-workgroup slabs.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bench9 store_bench9.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr size_t N = (size_t)8192 * 156 * 79;
__global__ void k_stride(float4 *out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_slab(float4 *out, size_t n4) {
    size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    size_t b = (size_t)blockIdx.x * per, e = b + per < n4 ? b + per : n4;
    for (size_t i = b + threadIdx.x; i < e; i += blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_read(const float4 *in, size_t n4, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; acc += v.x + v.w; }
    if (acc == 123.456f) *sink = acc;
}
int main() {
    float *bufs[10], *sink;
    CK(hipMalloc(&sink, 4));
    for (int k = 0; k < 10; k++) CK(hipMalloc(&bufs[k], N * 4));
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
    const size_t n4 = N / 4;
    timeit([&] { k_stride<<<256, 256>>>((float4 *)bufs[0], n4); });
    printf("buffer        :"); for (int k = 0; k < 10; k++) printf(" %6d", k); printf("\n");
    printf("fill 256x256  :"); for (int k = 0; k < 10; k++) printf(" %6.1f", 1e3 * timeit([&] { k_stride<<<256, 256>>>((float4 *)bufs[k], n4); })); printf("  us\n");
    printf("slabs 256x256 :"); for (int k = 0; k < 10; k++) printf(" %6.1f", 1e3 * timeit([&] { k_slab<<<256, 256>>>((float4 *)bufs[k], n4); })); printf("  us\n");
    printf("read 2048x256 :"); for (int k = 0; k < 10; k++) printf(" %6.1f", 1e3 * timeit([&] { k_read<<<2048, 256>>>((const float4 *)bufs[k], n4, sink); })); printf("  us\n");
    return 0;
}

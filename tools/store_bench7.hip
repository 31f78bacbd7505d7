// Does the size of the address window the whole chip writes into at one time matter?  256 x 256 threads; the buffer is
// processed in super-blocks of W bytes; inside a super-block every workgroup owns one contiguous slab of W / 256 bytes
// (float4 per lane, block-linear).  W = whole buffer is the plain slab pattern; small W approaches the global wavefront.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bench7 store_bench7.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr size_t N = (size_t)8192 * 156 * 79;

__global__ __launch_bounds__(256) void k_window(float4 *out, size_t n4, size_t win4) {
    const size_t slab4 = win4 / gridDim.x;
    for (size_t base = 0; base < n4; base += win4) {
        const size_t b = base + (size_t)blockIdx.x * slab4;
        size_t e = b + slab4;
        if (e > n4) e = n4;
        for (size_t i = b + threadIdx.x; i < e; i += blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

int main() {
    float *out;
    CK(hipMalloc(&out, N * 4 + (64 << 20)));
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
    for (int rep = 0; rep < 2; rep++)
        for (size_t wmb : {1, 2, 4, 8, 16, 32, 64, 128, 256, 512}) {
            const size_t win4 = (wmb << 20) / 16;
            float ms = timeit([&] { k_window<<<256, 256>>>((float4 *)out, n4, win4); });
            printf("window %4zu MiB (slab %5zu KiB per workgroup)  %6.1f us  %7.1f GB/s\n", wmb, (wmb << 10) / 256, ms * 1e3, N * 4 / 1e9 / (ms * 1e-3));
            fflush(stdout);
        }
    return 0;
}

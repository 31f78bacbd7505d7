// Cache-policy bits on the store instruction (gfx950: sc0, sc1, nt) in the per-CU stream pattern: does any of them lift
// the 5.0-5.4 TB/s of private streams?  256 x 256 threads, each block walks its own contiguous slab, float4 per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bench6 store_bench6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr size_t N = (size_t)8192 * 156 * 79;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_slab(f4 *out, size_t n4) {
    size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    size_t b = (size_t)blockIdx.x * per, e = b + per < n4 ? b + per : n4;
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        f4 *p = out + i;
        if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
        if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
        if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
        if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    }
}

int main() {
    float *out;
    CK(hipMalloc(&out, N * 4 + 4096));
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
    const char *names[] = {"plain", "sc0", "sc1", "sc0 sc1", "nt", "sc0 sc1 nt", "sc1 nt"};
    for (int rep = 0; rep < 2; rep++) {
        float ms[7];
        ms[0] = timeit([&] { k_slab<0><<<256, 256>>>((f4 *)out, n4); });
        ms[1] = timeit([&] { k_slab<1><<<256, 256>>>((f4 *)out, n4); });
        ms[2] = timeit([&] { k_slab<2><<<256, 256>>>((f4 *)out, n4); });
        ms[3] = timeit([&] { k_slab<3><<<256, 256>>>((f4 *)out, n4); });
        ms[4] = timeit([&] { k_slab<4><<<256, 256>>>((f4 *)out, n4); });
        ms[5] = timeit([&] { k_slab<5><<<256, 256>>>((f4 *)out, n4); });
        ms[6] = timeit([&] { k_slab<6><<<256, 256>>>((f4 *)out, n4); });
        for (int m = 0; m < 7; m++) printf("block slab float4, store %-12s %6.1f us  %7.1f GB/s\n", names[m], ms[m] * 1e3, N * 4 / 1e9 / (ms[m] * 1e-3));
        fflush(stdout);
    }
    return 0;
}

// Is there an address -> XCD affinity for writes?  Workgroup w runs on XCD w % 8 (hardware dispatch rule).  A persistent
// grid of 256 x 256 threads writes 404 MB in pieces of P bytes; workgroup w only writes pieces q with
// q % 8 == (w % 8 + k) % 8, walking its pieces in address order (32 workgroups share one residue class and
// interleave inside it).  If some (P, k) is much faster than the others, writes have a home XCD at granularity P.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bench4 store_bench4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr size_t N = (size_t)8192 * 156 * 79;

__global__ __launch_bounds__(256) void k_affine(float4 *out, size_t n4, int piece_f4, int k, int scatter) {
    const int w = blockIdx.x, x = w & 7, j = w >> 3;            // XCD, index inside the XCD's 32 workgroups
    const size_t n_pieces = n4 / piece_f4;
    const int cls = (x + k) & 7;
    // pieces of this class: q = cls + 8 i; workgroup j takes i = j, j + 32, ... (scatter = 0) or a contiguous run (scatter = 1)
    const size_t n_cls = (n_pieces - cls + 7) / 8;
    size_t i0, i1, istep;
    if (!scatter) { i0 = j; i1 = n_cls; istep = 32; }
    else { const size_t per = (n_cls + 31) / 32; i0 = j * per; i1 = i0 + per < n_cls ? i0 + per : n_cls; istep = 1; }
    for (size_t i = i0; i < i1; i += istep) {
        float4 *p = out + (cls + 8 * i) * (size_t)piece_f4;
        for (int e = threadIdx.x; e < piece_f4; e += 256) p[e] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

int main() {
    float *out;
    CK(hipMalloc(&out, N * 4 + (1 << 20)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](std::function<void()> f) {
        for (int i = 0; i < 2; i++) f();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; i++) f();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 10;
    };
    const size_t n4 = N / 4;
    printf("base address %p (mod 1 MiB = %zu)\n", (void *)out, (size_t)out % (1 << 20));
    for (int scatter = 0; scatter < 2; scatter++)
        for (int pb : {1024, 2048, 4096, 8192, 16384, 65536, 262144}) {
            printf("piece %6d B, %s:", pb, scatter ? "runs " : "dealt");
            for (int k = 0; k < 8; k++) {
                float ms = timeit([&] { k_affine<<<256, 256>>>((float4 *)out, n4, pb / 16, k, scatter); });
                printf(" %5.1f", ms * 1e3);
            }
            printf("  us (k = 0..7)\n");
            fflush(stdout);
        }
    return 0;
}

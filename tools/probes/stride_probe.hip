// What does PHYSICALLY CONTIGUOUS memory (hipExtMallocWithFlags(.., hipDeviceMallocContiguous)) punish?  The frames kernels' store
// pattern runs 2.6-2.9x slower there than a fill (tools/probes/contiguous_alloc.py, slow_patterns): the address -> channel / bank map
// is a pure function of the offset in such a buffer, so a sweep over the SHAPE of the instantaneous write front shows its structure.
//   E1  2048 waves (256 workgroups x 8), wave w writes region w of R bytes front to back in 1 KB steps, then region w + 2048, ...:
//       at any instant the chip writes 2048 blocks of 1 KB that are R bytes apart.  R = 1 KB is a fill.
//   E2  the same with a skew: wave w starts its region at step (w * skew) mod (R / 1 KB) and wraps (fronts no longer aligned)
//   E3  R fixed (12 KB), step bytes per wave instruction 256 .. 1024 (16 .. 64 lanes x 16 B)
// usage: stride_probe [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));

__global__ __launch_bounds__(512) void streams(char *buf, size_t total, unsigned R, unsigned skew, unsigned step) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t nw = (size_t)gridDim.x * 8, gw = (size_t)blockIdx.x * 8 + wave;
    const unsigned nsteps = R / step, lanes = step / 16;
    const f4 v = {0.f, 0.f, 0.f, 0.f};
    for (size_t base = gw * R; base + R <= total; base += nw * R) {
        unsigned s = skew ? (unsigned)((gw * skew) % nsteps) : 0;
        for (unsigned k = 0; k < nsteps; k++) {
            if (lane < lanes) *(f4 *)(buf + base + (size_t)s * step + lane * 16) = v;
            s = s + 1 == nsteps ? 0 : s + 1;
        }
    }
}
// E4: the frames kernels' unit map (workgroup b: units 8b .. 8b+7, unit = tile x chunk, wave = candidate, two halves) with the candidate
// stride CS and the chunk-region stride RS as parameters; each region written as 12 steps of 1 KB from its base.
__global__ __launch_bounds__(512) void tiles(char *buf, unsigned CS, unsigned RS, int ntiles, int order) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 8; s++) {
        const int u = order == 0 ? blockIdx.x * 8 + s : s * gridDim.x + blockIdx.x;
        if (u >= ntiles * 4) break;
        const int tile = u / 4, chunk = order == 0 ? (u % 4 + blockIdx.x) % 4 : u % 4;
        for (int half = 0; half < 2; half++) {
            char *base = buf + ((size_t)tile * 16 + wave + 8 * half) * CS + (size_t)chunk * RS;
            for (int k = 0; k < 12; k++) *(f4u *)(base + k * 1024 + lane * 16) = v;
        }
    }
}
// E5: the DENSE layout (CS = 49 296, RS = 12 324: the product's), every region written as a head fragment up to the first multiple of A bytes,
// then 1 KB blocks from there (so every wave instruction after the head starts on a multiple of A), the last one clipped.
__global__ __launch_bounds__(512) void dense_aligned(char *buf, unsigned A, int ntiles) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 8; s++) {
        const int u = blockIdx.x * 8 + s;
        if (u >= ntiles * 4) break;
        const int tile = u / 4, chunk = (u % 4 + blockIdx.x) % 4;
        for (int half = 0; half < 2; half++) {
            const size_t b0 = ((size_t)tile * 16 + wave + 8 * half) * 49296 + (size_t)chunk * 12324;
            const size_t b = (b0 + 15) & ~(size_t)15, e = (b0 + 12324) & ~(size_t)15;     // 16-byte pieces inside the region
            const size_t a = (b + A - 1) / A * A;
            if (b + lane * 16 < a) *(f4 *)(buf + b + lane * 16) = v;                       // head fragment (A <= 1024)
            for (size_t p = a + lane * 16; p < e; p += 1024) *(f4 *)(buf + p) = v;
        }
    }
}
// E6: the dense layout again, but a region OWNS whole G-byte granules: it covers [rd(b0), rd(b0 + 12 324)) with rd = round down to a multiple of G,
// so no granule is written by two workgroups and every wave instruction starts on a multiple of G.
__global__ __launch_bounds__(512) void dense_owned(char *buf, unsigned G, int ntiles) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 8; s++) {
        const int u = blockIdx.x * 8 + s;
        if (u >= ntiles * 4) break;
        const int tile = u / 4, chunk = (u % 4 + blockIdx.x) % 4;
        for (int half = 0; half < 2; half++) {
            const size_t b0 = ((size_t)tile * 16 + wave + 8 * half) * 49296 + (size_t)chunk * 12324;
            const size_t b = b0 / G * G, e = (b0 + 12324) / G * G;
            for (size_t p = b + lane * 16; p < e; p += 1024) *(f4 *)(buf + p) = v;
        }
    }
}
static float run_owned(char *buf, unsigned G, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) dense_owned<<<256, 512>>>(buf, G, 512);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) dense_owned<<<256, 512>>>(buf, G, 512);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}
static float run_dense(char *buf, unsigned A, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) dense_aligned<<<256, 512>>>(buf, A, 512);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) dense_aligned<<<256, 512>>>(buf, A, 512);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}
static float run_tiles(char *buf, unsigned CS, unsigned RS, int order, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) tiles<<<256, 512>>>(buf, CS, RS, 512, order);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) tiles<<<256, 512>>>(buf, CS, RS, 512, order);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}
static float run(char *buf, size_t total, unsigned R, unsigned skew, unsigned step, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) streams<<<256, 512>>>(buf, total, R, skew, step);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) streams<<<256, 512>>>(buf, total, R, skew, step);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}
int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    const size_t total = (size_t)8192 * 156 * 79 * 4;
    char *plain[4] = {nullptr, nullptr, nullptr, nullptr}, *contig = nullptr;
    const size_t alloc = (size_t)8192 * 54000;   // E4 varies the candidate stride up to 53 248 bytes: 8191 * 53 248 + 4 * 12 416 < alloc
    for (int i = 0; i < 4; i++) CK(hipMalloc((void **)&plain[i], alloc));
    CK(hipExtMallocWithFlags((void **)&contig, alloc, hipDeviceMallocContiguous));
    printf("E1: region bytes R (front = 2048 x 1 KB, R apart)      plain0   plain3   contiguous   (us per 404 MB)\n");
    const unsigned Rs[] = {1024, 2048, 3072, 4096, 5120, 6144, 8192, 10240, 12288, 16384, 20480, 24576, 32768, 49152, 65536, 98304, 131072, 196608};
    for (unsigned R : Rs)
        printf("  R = %7u                                          %7.1f  %7.1f  %7.1f\n", R, run(plain[0], total, R, 0, 1024, reps), run(plain[3], total, R, 0, 1024, reps), run(contig, total, R, 0, 1024, reps));
    printf("E2: R = 12288, wave w starts at step (w * skew) mod 12\n");
    for (unsigned skew : {0u, 1u, 5u, 7u})
        printf("  skew = %u                                             %7.1f  %7.1f  %7.1f\n", skew, run(plain[0], total, 12288, skew, 1024, reps), run(plain[3], total, 12288, skew, 1024, reps), run(contig, total, 12288, skew, 1024, reps));
    printf("E2b: R = 49152, skew\n");
    for (unsigned skew : {0u, 1u, 7u, 13u})
        printf("  skew = %2u                                            %7.1f  %7.1f  %7.1f\n", skew, run(plain[0], total, 49152, skew, 1024, reps), run(plain[3], total, 49152, skew, 1024, reps), run(contig, total, 49152, skew, 1024, reps));
    printf("E3: R = 12288, bytes per wave instruction\n");
    for (unsigned step : {256u, 512u, 1024u})
        printf("  step = %4u                                          %7.1f  %7.1f  %7.1f\n", step, run(plain[0], total, 12288, 0, step, reps), run(plain[3], total, 12288, 0, step, reps), run(contig, total, 12288, 0, step, reps));
    printf("E4: the kernels' unit map, 8192 candidates x 4 regions x 12 KB written (403 MB); CS = candidate stride, RS = region stride\n");
    const unsigned cs_rs[][2] = {{49296, 12324}, {49152, 12288}, {49280, 12320}, {49408, 12352}, {49664, 12416}, {49296, 12288}, {49152 + 1024, 12288}, {49152 + 4096, 12288}};
    for (auto &c : cs_rs)
        for (int order = 0; order < 2; order++) {
            if ((size_t)8191 * c[0] + 3 * (size_t)c[1] + 12288 > alloc || 3 * c[1] + 12288 > c[0]) { printf("  CS = %u RS = %u skipped (out of bounds)\n", c[0], c[1]); continue; }
            printf("  CS = %5u RS = %5u %s                  %7.1f  %7.1f  %7.1f\n", c[0], c[1], order ? "units round robin " : "8 units per group  ", run_tiles(plain[0], c[0], c[1], order, reps),
                   run_tiles(plain[3], c[0], c[1], order, reps), run_tiles(contig, c[0], c[1], order, reps));
        }
    printf("E5: dense layout (CS = 49296, RS = 12324), wave instructions start on multiples of A bytes\n");
    for (unsigned A : {16u, 32u, 64u, 128u, 256u, 512u, 1024u})
        printf("  A = %4u                                             %7.1f  %7.1f  %7.1f\n", A, run_dense(plain[0], A, reps), run_dense(plain[3], A, reps), run_dense(contig, A, reps));
    printf("E6: dense layout, a region owns whole granules of G bytes (no granule written by two workgroups)\n");
    for (unsigned G : {16u, 32u, 64u, 128u, 256u, 1024u})
        printf("  G = %4u                                             %7.1f  %7.1f  %7.1f\n", G, run_owned(plain[0], G, reps), run_owned(plain[3], G, reps), run_owned(contig, G, reps));
    return 0;
}

// Which store patterns does slow-class memory punish?  Allocates plain 404 MB buffers until it holds one of each class (pattern /
// fill ratio above / below 1.15, the library's criterion), then times variants of the sweep's store stream on both:
//   0  the replica of the tile-major order (mg_placement_pattern_kernel): a wave alternates between its two candidates per trip
//   1  the same, a wave finishes one candidate's 39 rows before it starts the other's
//   2  chunk-stationary order: a workgroup keeps one chunk and walks 8 tiles
//   3  16 waves per workgroup, one candidate each
//   4  the candidate-chunk regions written as aligned 1 KB blocks (what a fill would do inside the same regions, same order of regions)
//   5  tile-major, but a wave writes its candidate's WHOLE row (4 chunks = 49 KB contiguous) before the workgroup moves on
//   6  a plain fill
//   7  units dealt round robin: at any time the chip writes one compact window of 64 tiles (50 MB) that moves through the buffer
//   8  wave-level regions in address order: the 2048 waves write 2048 consecutive (candidate, chunk) regions = one contiguous 25 MB window
//  10  chunk-stationary order with the four chunk workgroups of a tile block on one XCD (round 5); 11: that map writing owned 128 B granules
//   9  the same with 4 KB per wave at a time (the 2048 waves write one contiguous 8 MB window: a fill with this kernel's instruction shape)
// usage: slow_patterns [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));
#define T 156
#define D 79
#define NF 39
#define NCH 4
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void rows3(float *out, size_t cand, int chunk, int f0, int lane) {
    const int fsub = lane / 20, ql = lane % 20, f = f0 + fsub;
    if (lane < 60 && f < NF) {
        float *p = out + (cand * T + (size_t)(chunk * NF + f)) * D + (ql == 19 ? 75 : 4 * ql);
        const f4u v = {0.f, 0.f, 0.f, 0.f};
        *(f4u *)p = v;
    }
}
template <int MODE>
__global__ __launch_bounds__(1024) void pattern(float *out, int ntiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int U = ntiles * NCH, per = (U + (int)gridDim.x - 1) / (int)gridDim.x;
    if (MODE == 2) {   // chunk-stationary: workgroup w keeps chunk w % 4, walks the tiles of block w / 4
        const int chunk = blockIdx.x % NCH, q = blockIdx.x / NCH, nq = gridDim.x / NCH, tper = (ntiles + nq - 1) / nq;
        for (int s = 0; s < tper; s++) {
            const int tile = q * tper + s;
            if (tile >= ntiles) break;
            for (int f0 = 0; f0 < NF; f0 += 3)
                for (int half = 0; half < 2; half++) rows3(out, (size_t)tile * 16 + wave + 8 * half, chunk, f0, lane);
        }
        return;
    }
    if (MODE == 10 || MODE == 11) {   // chunk-stationary, the four chunk workgroups of a tile block on ONE XCD (workgroups go to XCD blockIdx % 8)
        const int x = blockIdx.x % 8, r = blockIdx.x / 8, chunk = r % NCH, q = x + 8 * (r / NCH), nq = gridDim.x / NCH, tper = (ntiles + nq - 1) / nq;
        for (int s = 0; s < tper; s++) {
            const int tile = q * tper + s;
            if (tile >= ntiles) break;
            if (MODE == 10) {
                for (int f0 = 0; f0 < NF; f0 += 3)
                    for (int half = 0; half < 2; half++) rows3(out, (size_t)tile * 16 + wave + 8 * half, chunk, f0, lane);
            } else {   // 11: the same map, regions as granule-owning aligned blocks (what the map can reach at best)
                for (int half = 0; half < 2; half++) {
                    const size_t b0 = (((size_t)tile * 16 + wave + 8 * half) * T + (size_t)chunk * NF) * D * 4;
                    const size_t b = b0 / 128 * 128, e = (b0 + (size_t)NF * D * 4) / 128 * 128;
                    for (size_t p = b + lane * 16; p < e; p += 1024) { const f4 v = {0.f, 0.f, 0.f, 0.f}; *(f4 *)((char *)out + p) = v; }
                }
            }
        }
        return;
    }
    if (MODE == 5) {   // whole rows: workgroup w takes tiles, a wave writes 4 chunks of candidate A, then of candidate B
        const int tper = (ntiles + gridDim.x - 1) / gridDim.x;
        for (int s = 0; s < tper; s++) {
            const int tile = blockIdx.x * tper + s;
            if (tile >= ntiles) break;
            for (int half = 0; half < 2; half++)
                for (int chunk = 0; chunk < NCH; chunk++)
                    for (int f0 = 0; f0 < NF; f0 += 3) rows3(out, (size_t)tile * 16 + wave + 8 * half, chunk, f0, lane);
        }
        return;
    }
    if (MODE == 7) {
        for (int s = 0; s < per; s++) {
            const int u = s * gridDim.x + blockIdx.x;
            if (u >= U) break;
            const int tile = u / NCH, chunk = u % NCH;
            for (int f0 = 0; f0 < NF; f0 += 3)
                for (int half = 0; half < 2; half++) rows3(out, (size_t)tile * 16 + wave + 8 * half, chunk, f0, lane);
        }
        return;
    }
    if (MODE == 8) {
        const int nw = gridDim.x * 8, gw = blockIdx.x * 8 + wave, R = ntiles * 16 * NCH;
        for (int r = gw; r < R; r += nw)
            for (int f0 = 0; f0 < NF; f0 += 3) rows3(out, (size_t)(r / NCH), r % NCH, f0, lane);
        return;
    }
    if (MODE == 9) {
        const size_t nw = (size_t)gridDim.x * 8, gw = (size_t)blockIdx.x * 8 + wave, total = (size_t)ntiles * 16 * T * D * 4;
        for (size_t base = gw * 4096; base + 4096 <= total; base += nw * 4096)
            for (int k = 0; k < 4; k++) { const f4 v = {0.f, 0.f, 0.f, 0.f}; *(f4 *)((char *)out + base + k * 1024 + lane * 16) = v; }
        return;
    }
    for (int s = 0; s < per; s++) {
        const int u = blockIdx.x * per + s;
        if (u >= U) break;
        const int tile = u / NCH, chunk = (u % NCH + blockIdx.x) % NCH;
        if (MODE == 0) {
            for (int f0 = 0; f0 < NF; f0 += 3)
                for (int half = 0; half < 2; half++) rows3(out, (size_t)tile * 16 + wave + 8 * half, chunk, f0, lane);
        } else if (MODE == 1) {
            for (int half = 0; half < 2; half++)
                for (int f0 = 0; f0 < NF; f0 += 3) rows3(out, (size_t)tile * 16 + wave + 8 * half, chunk, f0, lane);
        } else if (MODE == 3) {
            for (int f0 = 0; f0 < NF; f0 += 3) rows3(out, (size_t)tile * 16 + wave, chunk, f0, lane);
        } else if (MODE == 4) {
            for (int half = 0; half < 2; half++) {
                const size_t cand = (size_t)tile * 16 + wave + 8 * half;
                char *b = (char *)(out + (cand * T + (size_t)chunk * NF) * D), *e = b + (size_t)NF * D * 4;
                char *a = (char *)(((size_t)b + 15) & ~(size_t)15);
                for (char *p = a + lane * 16; p + 16 <= e; p += 1024) { const f4 v = {0.f, 0.f, 0.f, 0.f}; *(f4 *)p = v; }
            }
        }
    }
}
__global__ __launch_bounds__(256) void fill(f4 *buf, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const f4 v = {0.f, 0.f, 0.f, 0.f}; buf[i] = v; }
}

static float time_mode(int mode, float *buf, int ntiles, size_t bytes, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&] {
        switch (mode) {
            case 0: pattern<0><<<256, 512>>>(buf, ntiles); break;
            case 1: pattern<1><<<256, 512>>>(buf, ntiles); break;
            case 2: pattern<2><<<256, 512>>>(buf, ntiles); break;
            case 3: pattern<3><<<256, 1024>>>(buf, ntiles); break;
            case 4: pattern<4><<<256, 512>>>(buf, ntiles); break;
            case 5: pattern<5><<<256, 512>>>(buf, ntiles); break;
            case 7: pattern<7><<<256, 512>>>(buf, ntiles); break;
            case 8: pattern<8><<<256, 512>>>(buf, ntiles); break;
            case 9: pattern<9><<<256, 512>>>(buf, ntiles); break;
            case 10: pattern<10><<<256, 512>>>(buf, ntiles); break;
            case 11: pattern<11><<<256, 512>>>(buf, ntiles); break;
            default: fill<<<(unsigned)((bytes / 16 + 255) / 256), 256>>>((f4 *)buf, bytes / 16); break;
        }
    };
    for (int i = 0; i < 3; i++) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return 1e3f * ms / reps;
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20, ntiles = 512;
    const size_t bytes = (size_t)ntiles * 16 * T * D * 4;
    std::vector<float *> held;
    float *fast = nullptr, *slow = nullptr;
    for (int i = 0; i < 40 && !(fast && slow); i++) {
        float *p = nullptr;
        if (hipMalloc((void **)&p, bytes) != hipSuccess) break;
        const float tf = time_mode(6, p, ntiles, bytes, 6), tp = time_mode(0, p, ntiles, bytes, 6);
        const bool is_slow = tp / tf > 1.15f;
        if (is_slow && !slow) slow = p;
        else if (!is_slow && !fast) fast = p;
        else held.push_back(p);
    }
    const char *names[12] = {"tile-major replica (two candidates alternating)", "... one candidate after the other", "chunk-stationary order",
                            "16 waves, one candidate each", "regions as aligned 1 KB blocks", "whole rows per wave (49 KB contiguous)", "plain fill",
                            "units round robin (50 MB window)", "wave regions in address order (25 MB window)", "4 KB per wave in address order (8 MB window)",
                            "chunk-stationary, a tile block's 4 chunks on one XCD", "... the same map, regions as owned 128 B granules"};
    float *contig = nullptr;   // physically contiguous memory (round 5): the slowest class of all for the pattern, at the fill's rate for a fill
    if (hipExtMallocWithFlags((void **)&contig, bytes, hipDeviceMallocContiguous) != hipSuccess) contig = nullptr;
    printf("%-52s %10s %10s %10s\n", "us per launch", fast ? "fast buf" : "(none)", slow ? "slow buf" : "(none)", contig ? "contiguous" : "(none)");
    for (int m = 0; m < 12; m++) {
        const float a = fast ? time_mode(m, fast, ntiles, bytes, reps) : 0.f, b = slow ? time_mode(m, slow, ntiles, bytes, reps) : 0.f;
        const float c = contig ? time_mode(m, contig, ntiles, bytes, reps) : 0.f;
        printf("%-52s %10.1f %10.1f %10.1f\n", names[m], a, b, c);
    }
    return 0;
}

// Shared pieces of the LDS-staged frames kernels (gfx950): launch arguments, the diagnostic build's switches and timers,
// the LDS progress-counter hand-off, unit cursors, the quad-row sweep's tap and store helpers, the fused mixture scoring.
// Included by mg_frames_ws.hip (tile-major), mg_frames_cs.hip (chunk-stationary) and mg_frames.hip (planning and dispatch).
#pragma once
#include <cstdio>
#include <cstdlib>

#include "mg_internal.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include "mg_gmm_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte store
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

// scalars of one launch; the pointers are separate __restrict__ kernel parameters
struct mg_frames_args {
    int64_t B, ld;
    int32_t T, D, Dp, cshift, L, nroot, n_chunks, n_tiles, stride, max_wi, max_nt;
    int32_t debug;   // MG_DEBUG_FLAGS (ablations and timers, never set in production): 1 = producers idle, 2 = sweep idle,
                     // 4 = sweep stores only, 16 = per-wave phase timers, 32 = wave-0 sub-phases (serialising), 64 = row producers without MFMAs,
                     // 128 = no chunk rotation, 256 = no tile-round rotation, 512 = row producers without E' loads, 1024 = wave 0 idle
    int32_t nbuf;    // LDS ring depth (2 or 3)
    int32_t max_tiles;   // row tiles of the widest chunk window (chunk-stationary kernel: sizes its mean' window in LDS)
    int32_t rt_total;    // row tiles of the primitive's whole eigenvector image (the chunk-stationary kernel's quad copy lies behind the pair image: + rt_total * KK * 64 floats)
    int32_t gmm_staged;  // chunk-stationary kernel, fused mixture: its latent tiles, C-in rows and constants have room in LDS (staged at start-up, MG_CS_GMM_LDSX)
    int32_t cs_magic, cs_per, cs_rem;   // chunk-stationary kernel: workgroup w -> (chunk, block) without a division: q = (w * cs_magic) >> 20;
                                        // cs_per tiles per workgroup of a chunk, the first cs_rem one more
    mg_chunk ck[MG_ARG_CHUNKS];   // the first chunks' descriptors: read with the other arguments instead of a dependent trip to global memory
};

template <bool F64>
__device__ __forceinline__ double mg_load_lat(const void *lat, int64_t idx) {
    if (F64) return ((const double *)lat)[idx];
    return (double)((const float *)lat)[idx];
}

// Ablation switches and phase timers exist only in the diagnostic build (make libmg_hip_dbg.so, -DMG_DEBUG_BUILD):
// the product kernel carries none of their branches and never reads the environment.
#ifdef MG_DEBUG_BUILD
#define MG_DBG(bits) (a.debug & (bits))
// diagnostic phase timers (MG_DEBUG_FLAGS & 16): per-wave s_memtime deltas accumulated in
// registers over all units and written once at kernel end (a store inside the loop would put
// the producers' loads behind it in vmcnt order and distort what is being measured).
__device__ unsigned long long mg_dbg_wg[1024][2];      // [workgroup][begin, end] in 100 MHz ticks (first sweep wave)
__device__ unsigned long long mg_dbg_stamps[16][10];   // [wave][phase] of workgroup 0; [8] = shader cycles, [9] = 100 MHz ticks of the wave
__device__ unsigned long long mg_dbg_units[16][32][2];  // chunk-stationary kernel, workgroup 0: [wave][unit][work begins, work ends] in 100 MHz ticks
#define MG_UNIT_STAMP(u_, k_)                                                                                          \
    do {                                                                                                               \
        if ((a.debug & 16) && blockIdx.x == 0 && lane == 0 && (u_) < 32) mg_dbg_units[wave][u_][k_] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define MG_SUB_STAMP(row_, u_, k_)                                                                                     \
    do {                                                                                                               \
        if ((a.debug & 16) && blockIdx.x == 0 && lane == 0 && (u_) < 32) mg_dbg_units[row_][u_][k_] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define MG_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = __builtin_amdgcn_s_memtime(); \
    const unsigned long long st_c0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();
#define MG_STAMP(ph)                                                     \
    do {                                                                 \
        if (a.debug & 16) {                                              \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            st_acc[ph] += now_ - st_prev;                                \
            st_prev = now_;                                              \
        }                                                                \
    } while (0)
#define MG_STAMP_DUMP                                                    \
    do {                                                                 \
        if ((a.debug & 16) && blockIdx.x == 0 && lane == 0) {            \
            for (int ph_ = 0; ph_ < 8; ph_++) mg_dbg_stamps[wave][ph_] = st_acc[ph_]; \
            mg_dbg_stamps[wave][8] = __builtin_amdgcn_s_memtime() - st_c0; \
            mg_dbg_stamps[wave][9] = __builtin_amdgcn_s_memrealtime() - st_r0; \
        }                                                                \
        if ((a.debug & 16) && wave == 4 && lane == 0 && blockIdx.x < 1024) { \
            mg_dbg_wg[blockIdx.x][0] = st_r0;                            \
            mg_dbg_wg[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                \
    } while (0)

// Light stamps (MG_DEBUG_FLAGS & 32768, without 16): a handful of 100 MHz clock reads per wave kept in scalar registers and written ONCE when the wave
// ends -- no store inside the kernel's life, so the timeline is the product kernel's (the per-unit stamps above put global stores into the producers).
__device__ unsigned long long mg_dbg_lite[2][16][8];   // [workgroup 0 / workgroup 131][wave][stamp]
#define MG_LITE_DECL unsigned long long lite_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MG_LITE(i_) do { if (a.debug & 32768) lite_[i_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MG_LITE_DUMP                                                                                              \
    do {                                                                                                          \
        if ((a.debug & 32768) && (blockIdx.x == 0 || blockIdx.x == 131) && lane == 0)                             \
            for (int i_ = 0; i_ < 8; i_++) mg_dbg_lite[blockIdx.x ? 1 : 0][wave][i_] = lite_[i_];                 \
    } while (0)
extern "C" int mg_debug_dump_lite(void) {
    static unsigned long long h[2][16][8], zero[2][16][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(mg_dbg_lite), sizeof(h)) != hipSuccess) return -1;
    for (int w2 = 0; w2 < 2; w2++) {
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 12; w++) if (h[w2][w][0] && h[w2][w][0] < t0) t0 = h[w2][w][0];
        printf("light stamps, workgroup %d, us after its first wave's entry: [0] entry; producers (waves 0-3): [1] tail begins [3] terms done [4] log-sum-exp done;\n"
               "  [5] barrier passed [6] first unit begins (wave 0: its latent tile is published) [7] first unit published; sweep waves: [1] first sweep begins [2] last sweep begins [4] last sweep ends\n", w2 ? 131 : 0);
        for (int w = 0; w < 12; w++) {
            printf("  wave %2d:", w);
            for (int i = 0; i < 8; i++) printf(" %7.2f", h[w2][w][i] ? (h[w2][w][i] - t0) / 100.0 : -1.0);
            printf("\n");
        }
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_dbg_lite), zero, sizeof(zero));
    return 0;
}

extern "C" int mg_debug_dump_stamps(void) {
    unsigned long long h[16][10];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(mg_dbg_stamps), sizeof(h)) != hipSuccess) return -1;
    printf("per-wave cycles summed over the units of workgroup 0; phase p = time from stamp p-1 to stamp p\n");
    printf("  (0: loop top, 1: sweep: wait for producers, 2: row producers: tiles, 3: row producers: carried tiles / wave 0: root stage,\n"
           "   4: sweep / publish, 5: producers: wait for a free slot; sweep: publish)\n");
    for (int w = 0; w < 12; w++) {
        printf("wave %2d:", w);
        for (int ph = 0; ph < 8; ph++) printf(" %9llu", h[w][ph]);
        printf("  | %9llu cycles in %.2f us = %.3f GHz\n", h[w][8], h[w][9] / 100.0, h[w][9] ? h[w][8] / (h[w][9] * 10.0) : 0.0);
    }
    static unsigned long long wg[1024][2];
    if (hipMemcpyFromSymbol(wg, HIP_SYMBOL(mg_dbg_wg), sizeof(wg)) != hipSuccess) return -1;
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < 1024; i++) if (wg[i][1] && wg[i][0] < t0) t0 = wg[i][0];
    printf("sweep wave 4 of every workgroup, us after the first one began: begin / end\n");
    for (int i = 0; i < 1024; i++) {
        if (!wg[i][1]) continue;
        if (i % 8 == 0) printf("\n  wg %3d:", i);
        printf(" %5.1f/%5.1f", (wg[i][0] - t0) / 100.0, (wg[i][1] - t0) / 100.0);
    }
    printf("\n");
    static unsigned long long un[16][32][2];
    if (hipMemcpyFromSymbol(un, HIP_SYMBOL(mg_dbg_units), sizeof(un)) != hipSuccess) return -1;
    unsigned long long u0 = ~0ull;
    for (int w = 0; w < 16; w++) if (un[w][0][0] && un[w][0][0] < u0) u0 = un[w][0][0];   // row 14: kernel entry / barrier passed (wave 0)
    if (u0 != ~0ull) {
        printf("chunk-stationary kernel, workgroup 0: per unit, us after the first stamp: work begins / ends\n");
        for (int w = 0; w < 16; w++) {   // rows 12..15: sub-phases of wave 0 (latents staged / root chains done, root image written / taps done)
            printf("wave %2d:", w);
            for (int u = 0; u < (w >= 14 ? 4 : 32); u++) if (w >= 14 || un[w][u][1] != 0) printf(" %5.1f/%5.1f", (un[w][u][0] - u0) / 100.0, (un[w][u][1] - u0) / 100.0);
            printf("\n");
        }
    }
    static unsigned long long zero[16][32][2];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_dbg_units), zero, sizeof(zero));
    return 0;
}
#else
#ifdef MG_DBG_CONST   // tools/build_variant.sh NAME -DMG_DBG_CONST=bits: an ablation compiled in, without the diagnostic build's stamps
#define MG_DBG(bits) ((MG_DBG_CONST) & (bits))
#else
#define MG_DBG(bits) 0
#endif
#define MG_LITE_DECL
#define MG_LITE(i_) do { } while (0)
#define MG_LITE_DUMP do { } while (0)
#define MG_STAMP_DECL
#define MG_STAMP(ph) do { } while (0)
#define MG_STAMP_DUMP do { } while (0)
#define MG_UNIT_STAMP(u_, k_) do { } while (0)
#define MG_SUB_STAMP(row_, u_, k_) do { } while (0)
#endif

// -----------------------------------------------------------------------------------------
// The hot-path kernel: persistent, wave-specialised.
//
// A unit = 16 candidates x one time chunk (consecutive samples whose spline taps fall in a
// window of <= 8 basis functions).  One workgroup per CU walks a contiguous run of units.
//
// Why the roles are split: (1) vmcnt retires in issue order, so a wave with stores in flight
// cannot consume a later load until the stores drain -- producers therefore only LOAD and
// consumers only STORE; (2) the contraction is L2-latency bound and the sweep is HBM bound:
// in one-shot workgroups they run in lockstep and add up instead of overlapping.
//
//   producer wave 0    : unit u's per-sample tables -> tb[slot]; unit u's root-translation rows by
//                        v_mfma_f64_16x16x4_f64 (C-in = mean') -> rs; their spline taps by a second f64
//                        MFMA (banded weight matrix) -> float32 root outputs ro[slot]
//   producer waves 1-3 : unit u's window of padded coefficient rows by v_mfma_f32_16x16x4_f32
//                        (A = E' fragments from L2, two tiles per round; B = the latent tile in
//                        registers; C-in = mean') -> buf[slot] [cand][i*Dp + d + cshift]; the row tiles the
//                        window shares with the previous chunk's are copied from the previous slot
//   consumer waves 4-11: the "quad-row" sweep of a finished unit: a wave owns two candidates; a lane owns 4
//                        consecutive channels of one sample (4 ds_read_b128 taps, 16 FMAs, one
//                        dwordx4 store), the last lane of each row group owns the root channels;
//                        64/20 samples per wave instruction, so one store instruction writes ~1 KB
//                        of consecutive bytes and the next continues where it ended.
//   No s_barrier in the unit loop: the slots form a ring of nbuf (3 when LDS allows, else 2) and the roles
//   hand units over through per-wave progress counters in LDS (mg_publish / mg_wait_*), so a slow consumer
//   wave delays only the recycling of its slot and the consumers' stores stay in flight throughout.
//   FUSE_GMM: after their last unit the producer waves score the workgroup's candidates against the mixture.
//
// LDS: buf[nbuf] = image [16][stride] f32; ro[nbuf] = root outputs [16][max_nt][4] f32; tb[nbuf] = w32
// [max_nt] float4 + image tap byte offsets [max_nt] int (max_nt = the grid's longest chunk, padded to 16); rs = float64 root image; prog = 32 counters
// (producer/consumer progress, mixture hand-off); FUSE_GMM: mixture terms and exponentials [2][K*16] f64 each.
// -----------------------------------------------------------------------------------------
#ifndef MG_SWEEP_LANEMAP
#define MG_SWEEP_LANEMAP 1   // the sweep's third sample in lanes 44 .. 63 (see the kernels)
#endif
#ifndef MG_SWEEP_FAST
#define MG_SWEEP_FAST 1      // the lean loop over a chunk's full trips (see the chunk-stationary kernel)
#endif
#ifndef MG_GMM_PAIR_LOADS
#define MG_GMM_PAIR_LOADS 0   // the fused mixture's component fragments two components per round of loads (A/B: tools/build_variant.sh)
#endif
#define MG_FUSE_MAX_KK 10   // fused mixture scoring: k-steps (4 latent components each) that fit the register budget
#define MG_WS_NPW 4      // producer waves
#define MG_WS_NCW 8      // consumer waves, two candidates each
#define MG_WS_BLOCK (64 * (MG_WS_NPW + MG_WS_NCW))

typedef int i32x4 __attribute__((ext_vector_type(4)));

// Producer/consumer hand-off through per-wave progress counters in LDS (no s_barrier in the unit loop):
// prog[w] = number of units wave w has finished.  A wave publishes after its own LDS traffic has completed
// (lgkmcnt(0)); LDS serves one wave's requests in order, so whoever sees the counter sees the data.
// (the counters are addressed through an explicit LDS pointer: a generic one becomes flat_load + vmcnt(0),
// which would drain the consumers' stores at every unit)
typedef __attribute__((address_space(3))) int mg_lds_int;
typedef __attribute__((address_space(3))) i32x4 mg_lds_i32x4;
__device__ __forceinline__ void mg_publish(mg_lds_int *prog, int wave, int lane, int done) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *(volatile mg_lds_int *)&prog[wave] = done;
}
// Workgroup barrier that orders LDS traffic only: vector-memory loads issued before it stay in flight across it
// (__syncthreads() carries a fence that drains them).
__device__ __forceinline__ void mg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void mg_wait_producers(const mg_lds_int *prog, int target) {   // waves 0..3
    for (;;) {
        const i32x4 v = *(const volatile mg_lds_i32x4 *)prog;
        const int m = min(min(v[0], v[1]), min(v[2], v[3]));
        if (__builtin_amdgcn_readfirstlane(m) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void mg_wait_row_producers(const mg_lds_int *prog, int target) {   // waves 1..3
    for (;;) {
        const i32x4 v = *(const volatile mg_lds_i32x4 *)prog;
        const int m = min(v[1], min(v[2], v[3]));
        if (__builtin_amdgcn_readfirstlane(m) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void mg_wait_consumers(const mg_lds_int *prog, int target) {   // waves 4..11
    for (;;) {
        const i32x4 v = *(const volatile mg_lds_i32x4 *)(prog + 4);
        const i32x4 x = *(const volatile mg_lds_i32x4 *)(prog + 8);
        const int m = min(min(min(v[0], v[1]), min(v[2], v[3])), min(min(x[0], x[1]), min(x[2], x[3])));
        if (__builtin_amdgcn_readfirstlane(m) >= target) break;
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
}

struct mg_unit {
    mg_chunk ck;
    int64_t b0;
    int ncand;
    int tile;
    int chunk;
};
struct mg_cursor {   // (tile, chunk) cursor over a workgroup's run of units: no division inside the unit loop
    int tile, chunk;
};
__device__ __forceinline__ mg_unit mg_unit_at(const mg_chunk *__restrict__ chunks, const mg_frames_args &a, const mg_cursor &c, int rot = 0) {
    mg_unit r;
    r.tile = c.tile;
    r.chunk = c.chunk + rot;
    if (r.chunk >= a.n_chunks) r.chunk -= a.n_chunks;
    r.ck = chunks[r.chunk];
    r.b0 = (int64_t)c.tile * MG_NCAND;
    r.ncand = (int)((a.B - r.b0) < MG_NCAND ? (a.B - r.b0) : MG_NCAND);
    return r;
}
__device__ __forceinline__ void mg_cursor_next(mg_cursor &c, int n_chunks) {
    if (++c.chunk == n_chunks) { c.chunk = 0; c.tile++; }
}

// 4 channels of one sample: taps are 4 consecutive basis rows of the image (byte pitch dp4).  DP4 > 0: the pitch is a
// compile-time constant, the three row offsets become immediate offsets of the LDS reads (no address arithmetic).
template <int DP4>
__device__ __forceinline__ f32x4 mg_quad_taps_t(const unsigned char *tp, const float4 w, int dp4_rt);
__device__ __forceinline__ f32x4 mg_quad_taps(const unsigned char *tp, const float4 w, int dp4) {
    const f32x4 t0 = *(const f32x4 *)tp;
    const f32x4 t1 = *(const f32x4 *)(tp + dp4);
    const f32x4 t2 = *(const f32x4 *)(tp + 2 * dp4);
    const f32x4 t3 = *(const f32x4 *)(tp + 3 * dp4);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float x = w.x * t0[e];
        x = fmaf(w.y, t1[e], x);
        x = fmaf(w.z, t2[e], x);
        x = fmaf(w.w, t3[e], x);
        v[e] = x;
    }
    return v;
}
template <int DP4>
__device__ __forceinline__ f32x4 mg_quad_taps_t(const unsigned char *tp, const float4 w, int dp4_rt) {
    return mg_quad_taps(tp, w, DP4 > 0 ? DP4 : dp4_rt);
}
// the same in two steps, so that a trip can request all of its 16 tap rows before the first FMA waits for any of them
struct mg_tap_rows { f32x4 t0, t1, t2, t3; };
template <int DP4>
__device__ __forceinline__ mg_tap_rows mg_quad_load(const unsigned char *tp, int dp4_rt) {
    const int dp4 = DP4 > 0 ? DP4 : dp4_rt;
    mg_tap_rows r;
    r.t0 = *(const f32x4 *)tp;
    r.t1 = *(const f32x4 *)(tp + dp4);
    r.t2 = *(const f32x4 *)(tp + 2 * dp4);
    r.t3 = *(const f32x4 *)(tp + 3 * dp4);
    return r;
}
__device__ __forceinline__ f32x4 mg_quad_fma(const mg_tap_rows &r, const float4 w) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float x = w.x * r.t0[e];
        x = fmaf(w.y, r.t1[e], x);
        x = fmaf(w.z, r.t2[e], x);
        x = fmaf(w.w, r.t3[e], x);
        v[e] = x;
    }
    return v;
}

// The mean/delta split's finish in the root lane of a row group: that lane's quad is the first quad of the padded rows,
// {padding..., root channels} = columns 0 .. 3 with root channel d in column cshift + d; v holds its four taps' sums (the deltas),
// mh / ml the sample's (Mhi, Mlo).  out[d] = Mhi[d] + (Mlo[d] + delta[d]) for d < nroot; the other elements are not stored
// (or, where every lane stores four floats, replaced by the row's borrowed channel 3).
struct mg_rootm { float4 a; float2 b; };   // {Mhi[0], Mhi[1], Mhi[2], Mlo[0]}, {Mlo[1], Mlo[2]}: one ds_read_b128 + one ds_read_b64
__device__ __forceinline__ mg_rootm mg_rootm_load(const float *tab, int f) {
    mg_rootm m;
    m.a = *(const float4 *)(tab + 8 * f);
    m.b = *(const float2 *)(tab + 8 * f + 4);
    return m;
}
template <int CSHIFT>
__device__ __forceinline__ f32x4 mg_root_finish(const f32x4 &v, const mg_rootm &m) {
    f32x4 o;
    o[0] = m.a.x + (m.a.w + v[CSHIFT]);
    o[1] = CSHIFT + 1 < 4 ? m.a.y + (m.b.x + v[CSHIFT + 1 < 4 ? CSHIFT + 1 : 3]) : 0.f;
    o[2] = CSHIFT + 2 < 4 ? m.a.z + (m.b.y + v[CSHIFT + 2 < 4 ? CSHIFT + 2 : 3]) : 0.f;
    o[3] = 0.f;
    return o;
}
__device__ __forceinline__ f32x4 mg_root_finish_rt(const f32x4 &v, const mg_rootm &m, int cshift) {
    return cshift == 1 ? mg_root_finish<1>(v, m) : cshift == 2 ? mg_root_finish<2>(v, m) : mg_root_finish<3>(v, m);
}

// all active lanes store four floats at base (wave-uniform) + a 32-bit byte offset: one store instruction, no lane classes
__device__ __forceinline__ void mg_store4_at(float *base, unsigned byte_off, const f32x4 &v) {
    *(f32x4u *)((char *)base + byte_off) = v;
}

// the same with the base in scalar registers and the lane's part as the instruction's 32-bit offset: no 64-bit address per lane
__device__ __forceinline__ void mg_store4_s(float *base, unsigned byte_off, const f32x4 &v) {
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
}
// The root lane of a row group after the taps: it ran them on the first quad lane's columns, so v[0] is the row's channel 3;
// its own three channels come from wave 0's root outputs {r0, r1, r2, -}.  Stored: {r0, r1, r2, channel 3}.
__device__ __forceinline__ f32x4 mg_root_merge(const f32x4 &v, const float *ro) {
    const float c3 = v[0];
    f32x4 o = *(const f32x4 *)ro;
    o[3] = c3;
    return o;
}

__device__ __forceinline__ void mg_store_n(float *op, const f32x4 &v, int n) {
    if (n == 4) {
        *(f32x4u *)op = v;
    } else if (n == 3) {
        f32x3u t = {v[0], v[1], v[2]};
        *(f32x3u *)op = t;
    } else if (n == 2) {
        f32x2u t = {v[0], v[1]};
        *(f32x2u *)op = t;
    } else if (n == 1) {
        op[0] = v[0];
    }
}


// One wave's share of a unit's float32 coefficient window: row tiles pw, pw + npw, pw + 2 npw, ... of the
// window, each D = E'tile (16 x 4KK) . latent tile (4KK x 16) + mean' by KK chained v_mfma_f32_16x16x4_f32,
// written as four consecutive padded rows per lane (conflict-free since stride = 4 mod 32).
// Two tiles per round, the next two requested unconditionally (clamped) before the MFMAs of the current two
// issue: a conditional prefetch makes the compiler drain it with vmcnt(0) at the loop top.  (The compiler still
// folds the two register sets into one, so a round costs one L2 round trip + its MFMAs)
template <int KK>
__device__ __forceinline__ void mg_produce_f32(const float2 *__restrict__ ep, const float *__restrict__ mean32,
                                               const mg_chunk &ck, float *lds_c, int stride, int t_first, int pw, int npw,
                                               const float (&sfrag)[KK], int lane, int cl, int g, int rot = 0, int dbg = 0) {
    float2 fa[2][KK / 2], na[2][KK / 2];
    f32x4 fm[2], nm[2];
    auto load_tile = [&](int t, float2(&fr)[KK / 2], f32x4 &cin) {
        const int tc = t < ck.ntiles ? t : ck.ntiles - 1;   // clamp: redundant but in bounds
        const float2 *p = ep + ((size_t)(ck.rt0 + tc) * (KK / 2)) * 64 + lane;
#ifdef MG_DEBUG_BUILD
        if (dbg & 512) {   // ablation: no E' loads (with 32768 in the caller: on every other unit)
#pragma unroll
            for (int q = 0; q < KK / 2; q++) fr[q] = make_float2(0.5f, 0.25f);
            cin = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
#endif
#pragma unroll
        for (int q = 0; q < KK / 2; q++) fr[q] = p[q * 64];
        cin = *(const f32x4 *)(mean32 + (size_t)(ck.rt0 + tc) * 16 + 4 * g);
    };
    // rounds of two tiles; round r of this wave covers tiles t_first + pw + 2 npw r (+ npw); tiles below t_first are
    // carried over from the previous unit's window.  The rounds are walked from a workgroup-specific start (rot):
    // neighbouring workgroups then fetch different E' tiles at the same moment.
    const int span = ck.ntiles - t_first - pw;
    const int nrounds = span > 0 ? (span + 2 * npw - 1) / (2 * npw) : 0;
    if (nrounds == 0) return;
    int rr = rot % nrounds;
    int t = t_first + pw + 2 * npw * rr;
    load_tile(t, fa[0], fm[0]);
    load_tile(t + npw, fa[1], fm[1]);
    for (int r = 0; r < nrounds; r++) {
        if (++rr == nrounds) rr = 0;
        const int tn = t_first + pw + 2 * npw * rr;
        load_tile(tn, na[0], nm[0]);
        load_tile(tn + npw, na[1], nm[1]);
        f32x4 acc0 = fm[0], acc1 = fm[1];
#ifdef MG_DEBUG_BUILD
        if (dbg & 64) {   // ablation: no MFMAs (the loaded fragments stay live)
#pragma unroll
            for (int q = 0; q < KK / 2; q++) { asm volatile("" :: "v"(fa[0][q].x), "v"(fa[0][q].y), "v"(fa[1][q].x), "v"(fa[1][q].y)); }
        } else
#endif
#pragma unroll
        for (int q = 0; q < KK / 2; q++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0][q].x, sfrag[2 * q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1][q].x, sfrag[2 * q], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0][q].y, sfrag[2 * q + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1][q].y, sfrag[2 * q + 1], acc1, 0, 0, 0);
        }
        // D[row = 4g + reg][col = cl]: four consecutive padded rows of candidate cl
        if (t < ck.ntiles) *(f32x4 *)&lds_c[cl * stride + t * 16 + 4 * g] = acc0;
        if (t + npw < ck.ntiles) *(f32x4 *)&lds_c[cl * stride + (t + npw) * 16 + 4 * g] = acc1;
#pragma unroll
        for (int q = 0; q < KK / 2; q++) { fa[0][q] = na[0][q]; fa[1][q] = na[1][q]; }
        fm[0] = nm[0]; fm[1] = nm[1];
        t = tn;
    }
}

template <int KK, bool LAT_F64>
__device__ __forceinline__ void mg_load_sfrag(float (&sfrag)[KK], const void *lat, const mg_unit &un, int64_t ld, int L, int cl, int g) {
    typename mg_gmm_xt<LAT_F64>::type x[KK];
    mg_gmm_load_x<KK, LAT_F64>(x, lat, un.b0, un.ncand, ld, L, cl, g);
#pragma unroll
    for (int kk = 0; kk < KK; kk++) sfrag[kk] = (float)x[kk];
}


// The fused step kernel's mixture scoring: log p(s_b) for this workgroup's share of the candidates (at most two
// 16-candidate tiles: the launcher fuses only then), run by the four producer waves after their last unit, while
// the sweep waves drain the ring.  (Measured alternatives, all slower: by the sweep waves while the pipeline
// fills -- the float64 MFMAs delay the first unit; by the producers when the ring first fills, whole or one
// component per unit -- the mixture constants are read once per launch, miss L2 behind the store stream and
// each round trip stalls the producer long enough to starve the sweep.)
//  * mg_fused_gmm_terms: wave pw takes components pw, pw + 4, ... of both tiles (terms -> LDS); the loads of
//    a component and of both latent tiles are all issued before the first MFMA (two components per round do
//    not fit the register budget).
//  * mg_fused_gmm_finish: wave b (0, 1) finishes tile b (log-sum-exp -> logp) once gdone[0..3] say that all
//    terms are written.
template <int KK, bool LAT_F64>
__device__ __forceinline__ void mg_fused_gmm_terms(mg_lds_int *prog, const double *__restrict__ gPpack,
                                                   const double *__restrict__ gmP, const double *__restrict__ gcst,
                                                   const void *__restrict__ lat, int64_t B, int64_t ld, int L, int n_tiles,
                                                   int gK, int gJT, int pw, int lane, int group) {
    const int cl = lane & 15, g = lane >> 4;
    mg_lds_f64 *gterms = (mg_lds_f64 *)(prog + 32);   // [2][K*16]
    const int64_t gt0 = (int64_t)blockIdx.x * n_tiles / gridDim.x + 2 * group;   // this group's (at most two) tiles
    const int64_t gt1 = ((int64_t)blockIdx.x + 1) * n_tiles / gridDim.x;
    if (gt0 < gt1) {
        const bool has_b = gt0 + 1 < gt1;
        typename mg_gmm_xt<LAT_F64>::type xa[KK], xb[KK];
        {
            const int64_t ba = gt0 * MG_NCAND, bb = (has_b ? gt0 + 1 : gt0) * MG_NCAND;
            const int na = (int)((B - ba) < MG_NCAND ? (B - ba) : MG_NCAND);
            const int nb = (int)((B - bb) < MG_NCAND ? (B - bb) : MG_NCAND);
            mg_gmm_load_x<KK, LAT_F64>(xa, lat, ba, na, ld, L, cl, g);
            mg_gmm_load_x<KK, LAT_F64>(xb, lat, bb, nb, ld, L, cl, g);
        }
#if MG_GMM_PAIR_LOADS
        // two components' fragments requested together: in the tail every load waits behind the store stream's queue, so a round of
        // loads is a round trip of microseconds -- one round for a wave's two components (K = 8) instead of two
        for (int k = pw; k < gK; k += 2 * MG_WS_NPW) {
            const int k2 = k + MG_WS_NPW;
            mg_gmm_frag<KK> f, f2;
            mg_gmm_load_component<KK>(f, gPpack, gmP, gcst, k, gJT, lane, cl, gK);
            if (k2 < gK) mg_gmm_load_component<KK>(f2, gPpack, gmP, gcst, k2, gJT, lane, cl, gK);
            typedef typename mg_gmm_xt<LAT_F64>::type XT;
            mg_gmm_apply_component<KK, XT, true>(f, k, gJT, xa, gterms, cl, g);
            if (has_b) mg_gmm_apply_component<KK, XT, true>(f, k, gJT, xb, gterms + gK * 16, cl, g);
            if (k2 < gK) {
                mg_gmm_apply_component<KK, XT, true>(f2, k2, gJT, xa, gterms, cl, g);
                if (has_b) mg_gmm_apply_component<KK, XT, true>(f2, k2, gJT, xb, gterms + gK * 16, cl, g);
            }
        }
#else
        for (int k = pw; k < gK; k += MG_WS_NPW) {
            mg_gmm_frag<KK> f;
            mg_gmm_load_component<KK>(f, gPpack, gmP, gcst, k, gJT, lane, cl, gK);
            mg_gmm_apply_component(f, k, gJT, xa, gterms, cl, g);
            if (has_b) mg_gmm_apply_component(f, k, gJT, xb, gterms + gK * 16, cl, g);
        }
#endif
    }
    mg_publish(prog + 16, pw, lane, group + 1);   // gdone[pw]
}

// The chunk-stationary kernel's form for float32 latents (round 5): the group's two latent tiles wait in LDS since start-up (gx, [2][KK][64]
// float32 A fragments; mpl, cstl: the components' C-in rows and constants -- all staged by the sweep waves that produce nothing), so a producer wave has the registers to request BOTH of its
// components at once -- one round trip through a memory pipe that the last unit's stores keep full (~5 us each, measured: the tail's two
// dependent rounds were 13.3 us) instead of two.  Same components per wave, same arithmetic per (tile, component): the same bits.
template <int KK>
__device__ __forceinline__ void mg_fused_gmm_terms_ldsx(mg_lds_int *prog, const double *__restrict__ gPpack, const mg_lds_f32 *gx, const mg_lds_f64 *mpl,
                                                        const mg_lds_f64 *cstl, int n_tiles, int gK, int gJT, int pw, int lane, bool early) {
    const int cl = lane & 15, g = lane >> 4;
    mg_lds_f64 *gterms = (mg_lds_f64 *)(prog + 32);   // [2][K*16]
    const int64_t gt0 = (int64_t)blockIdx.x * n_tiles / gridDim.x;
    const int64_t gt1 = ((int64_t)blockIdx.x + 1) * n_tiles / gridDim.x;
    if (gt0 < gt1) {
        const bool has_b = gt0 + 1 < gt1;
        // this wave's components: pw, pw + 4, pw + 8, pw + 12 -- without pw + 4 where the start-up has scored components 4 .. 7 already (early)
        int ks[4], nk = 0;
        for (int k = pw; k < gK; k += MG_WS_NPW)
            if (!(early && k >= MG_WS_NPW && k < 2 * MG_WS_NPW)) ks[nk++] = k;
        for (int i = 0; i < nk; i += 2) {
            const int k = ks[i], k2 = i + 1 < nk ? ks[i + 1] : gK;
            mg_gmm_frag<KK> f, f2;
            mg_gmm_load_pf2<KK>(f, gPpack, gK, k, gJT, lane);
            if (k2 < gK) mg_gmm_load_pf2<KK>(f2, gPpack, gK, k2, gJT, lane);
            mg_gmm_apply_component_ldsx<KK>(f, k, gJT, gx, lane, mpl, cstl, gterms, cl, g);
            if (has_b) mg_gmm_apply_component_ldsx<KK>(f, k, gJT, gx + KK * 64, lane, mpl, cstl, gterms + gK * 16, cl, g);
            if (k2 < gK) {
                mg_gmm_apply_component_ldsx<KK>(f2, k2, gJT, gx, lane, mpl, cstl, gterms, cl, g);
                if (has_b) mg_gmm_apply_component_ldsx<KK>(f2, k2, gJT, gx + KK * 64, lane, mpl, cstl, gterms + gK * 16, cl, g);
            }
        }
    }
    mg_publish(prog + 16, pw, lane, 1);   // gdone[pw]
}

// Start-up half of the staged form (MG_CS_GMM_EARLY_HALF): sweep wave 4 + MG_CS_NSP + i, idle until the first unit is in LDS, scores component 4 + i on the
// workgroup's two latent tiles -- its fragments requested before the start-up barrier, the matrix pipes still idle (the row producers wait for their
// eigenvector fragments) -- and leaves the terms where the tail's log-sum-exp finds them.  Same device code per (tile, component): the same bits.
template <int KK>
__device__ __forceinline__ void mg_fused_gmm_early_component(mg_lds_int *prog, const mg_gmm_frag<KK> &f, int k, const mg_lds_f32 *gx, const mg_lds_f64 *mpl,
                                                             const mg_lds_f64 *cstl, int n_tiles, int gK, int gJT, int lane) {
    const int cl = lane & 15, g = lane >> 4;
    mg_lds_f64 *gterms = (mg_lds_f64 *)(prog + 32);   // [2][K*16]
    const int64_t gt0 = (int64_t)blockIdx.x * n_tiles / gridDim.x;
    const int64_t gt1 = ((int64_t)blockIdx.x + 1) * n_tiles / gridDim.x;
    mg_gmm_apply_component_ldsx<KK>(f, k, gJT, gx, lane, mpl, cstl, gterms, cl, g);
    if (gt0 + 1 < gt1) mg_gmm_apply_component_ldsx<KK>(f, k, gJT, gx + KK * 64, lane, mpl, cstl, gterms + gK * 16, cl, g);
}

__device__ __forceinline__ void mg_fused_gmm_finish(mg_lds_int *prog, float *__restrict__ logp, int64_t B, int n_tiles, int gK,
                                                    int pw, int lane, int group) {
    mg_lds_int *gdone = prog + 16;
    mg_lds_f64 *gterms = (mg_lds_f64 *)(prog + 32);   // [2][K*16]
    mg_lds_f64 *gexps = gterms + 2 * gK * 16;          // [2][K*16]
    const int64_t gt0 = (int64_t)blockIdx.x * n_tiles / gridDim.x + 2 * group;
    const int64_t gt1 = ((int64_t)blockIdx.x + 1) * n_tiles / gridDim.x;
    if (pw < 2 && gt0 + pw < gt1) {
        mg_wait_producers(gdone, group + 1);   // all four producer waves have written their components' terms
        const mg_lds_f64 *terms = gterms + pw * gK * 16;
        mg_lds_f64 *exps = gexps + pw * gK * 16;
        for (int e = lane; e < gK * 16; e += 64) exps[e] = mg_gmm_exp_entry(terms, gK, e);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t b0 = (gt0 + pw) * MG_NCAND;
        if (lane < MG_NCAND && b0 + lane < B) logp[b0 + lane] = (float)mg_gmm_logsumexp(terms, exps, gK, lane);
    }
    if (pw < 2) mg_publish(prog + 24, pw, lane, group + 1);   // gfin[pw]: the term buffer may be written again
}


struct mg_launch_events { hipEvent_t start = nullptr, stop = nullptr; };   // both NULL: an ordinary launch
// per-kernel launchers (each in its kernel's translation unit) and their dynamic-LDS attributes
int mg_launch_frames_ws(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a, bool lat_f64,
                        bool split, int buf_bytes, int lds, int grid, const mg_launch_events &ev);
int mg_launch_frames_cs(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a, bool lat_f64,
                        bool split, int buf_bytes, int lds, int grid, const mg_launch_events &ev);
int mg_frames_ws_attributes();
int mg_frames_cs_attributes();

// Back-projection kernels for gfx950 (MI355X):
//   frames[b][f][d] = sum_j w[f][j] * ( E'[(i0[f]+j) D + d] . s_b + mean'[(i0[f]+j) D + d] )
// replacing MotionPrimitive.back_project(s, False).get_motion_vector()
// (reference morphablegraphs/motion_model/motion_primitive.py:206-256 and
//  morphablegraphs/motion_model/motion_spline.py:71-92).
//
// f32 arithmetic contract (bit-exact CPU model: oracle/mg_oracle.c, *_f32model):
//   channels d >= nroot : c[r] = fmaf chain over k ascending starting from (float)mean'[r]
//       (== the v_mfma_f32_16x16x4_f32 accumulation order with C-in = mean),
//       out = w0*c0, fmaf(w1,c1,.), fmaf(w2,c2,.), fmaf(w3,c3,.)
//   channels d <  nroot : the same in float64 (v_mfma_f64_16x16x4_f64 / fma), out = (float)v64.
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
            for (int u = 0; u < 32 && (w >= 14 ? u < 4 : un[w][u][1] != 0); u++) printf(" %5.1f/%5.1f", (un[w][u][0] - u0) / 100.0, (un[w][u][1] - u0) / 100.0);
            printf("\n");
        }
    }
    static unsigned long long zero[16][32][2];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_dbg_units), zero, sizeof(zero));
    return 0;
}
#else
#define MG_DBG(bits) 0
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
#define MG_FUSE_MAX_KK 10   // fused mixture scoring: k-steps (4 latent components each) that fit the register budget
#define MG_WS_NPW 4      // producer waves
#define MG_WS_NCW 8      // consumer waves, two candidates each
#define MG_WS_BLOCK (64 * (MG_WS_NPW + MG_WS_NCW))
#define MG_TB_BYTES_N(nt) ((nt) * 16 + (nt) * 4)      // nt = the grid's longest chunk, rounded up to 16 samples
#define MG_RO_BYTES_N(nt) (MG_NCAND * (nt) * 16)

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

// all active lanes store four floats at base (wave-uniform) + a 32-bit byte offset: one store instruction, no lane classes
__device__ __forceinline__ void mg_store4_at(float *base, unsigned byte_off, const f32x4 &v) {
    *(f32x4u *)((char *)base + byte_off) = v;
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
        for (int k = pw; k < gK; k += MG_WS_NPW) {
            mg_gmm_frag<KK> f;
            mg_gmm_load_component<KK>(f, gPpack, gmP, gcst, k, gJT, lane, cl);
            mg_gmm_apply_component(f, k, gJT, xa, gterms, cl, g);
            if (has_b) mg_gmm_apply_component(f, k, gJT, xb, gterms + gK * 16, cl, g);
        }
    }
    mg_publish(prog + 16, pw, lane, group + 1);   // gdone[pw]
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

template <int KK, bool LAT_F64, bool FUSE_GMM>
__global__ __launch_bounds__(MG_WS_BLOCK) void mg_frames_ws_kernel(
    const float *__restrict__ Epack,      // [RT][KK/2][64][2]
    const float *__restrict__ mean32,     // [RT*16]
    const double *__restrict__ Erpack,    // [RRT][KK][64]
    const double *__restrict__ meanroot,  // [RRT*16]
    const void *__restrict__ lat,         // (B, ld) f32 or f64
    const int32_t *__restrict__ i0tab,    // (T)
    const float4 *__restrict__ w32,       // (T)
    const double *__restrict__ wtap,      // [n_chunks][2][2][64] banded tap weights as f64 MFMA A fragments
    const mg_chunk *__restrict__ chunks,
    float *__restrict__ out,              // (B,T,D)
    const double *__restrict__ gPpack,    // FUSE_GMM: precision-Cholesky fragments [K][JT][KK][64]
    const double *__restrict__ gmP,       // FUSE_GMM: mu_k P_k [K][JT*16]
    const double *__restrict__ gcst,      // FUSE_GMM: per-component constants [K]
    float *__restrict__ logp,             // FUSE_GMM: (B) log p(s_b)
    const mg_frames_args a, const int gK, const int gJT, const int buf_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stride = a.stride, D = a.D, Dp = a.Dp, L = a.L, nroot = a.nroot;
    const int root_stride = a.max_wi * nroot + 1;
    const int nbuf = a.nbuf;
    const int max_nt = a.max_nt;
    const int MG_RO_BYTES = MG_RO_BYTES_N(max_nt), MG_TB_BYTES = MG_TB_BYTES_N(max_nt);
    unsigned char *ro_base = smem + nbuf * (size_t)buf_bytes;         // root outputs, one per ring slot
    unsigned char *tb_base = ro_base + nbuf * MG_RO_BYTES;            // per-sample tables, one per ring slot
    unsigned char *rs_base = tb_base + nbuf * MG_TB_BYTES;            // float64 root image (wave 0 only)
    mg_lds_int *prog = (mg_lds_int *)(rs_base + (size_t)MG_NCAND * root_stride * 8);   // 32 counters: progress, GMM
    if (tid < 32) prog[tid] = (tid == 26 || tid == 27) ? 0x7fffffff : 0;   // 26, 27: padding of the gfin wait
    __syncthreads();

    const int64_t U = (int64_t)a.n_tiles * a.n_chunks;
    const int64_t u_begin = (int64_t)blockIdx.x * U / gridDim.x;
    const int64_t u_end = ((int64_t)blockIdx.x + 1) * U / gridDim.x;
    const int n_units = (int)(u_end - u_begin);
    mg_cursor cur;
    cur.tile = (int)(u_begin / a.n_chunks);
    cur.chunk = (int)(u_begin - (int64_t)cur.tile * a.n_chunks);
    // When this workgroup owns whole tiles it walks each tile's chunks starting at chunk (blockIdx mod n_chunks): the
    // workgroups run nearly in lockstep, and without the rotation all 256 of them request the same E' rows from L2 at
    // the same time, the cold first unit above all (-2 % kernel time; MG_DEBUG_FLAGS & 128 switches it off).
    const int rot = (!MG_DBG(128) && cur.chunk == 0 && (n_units % a.n_chunks) == 0) ? (int)(blockIdx.x % a.n_chunks) : 0;
    const int cl = lane & 15, g = lane >> 4;

    if (wave >= MG_WS_NPW) {
        // ================= consumers =================
        const int cj = wave - MG_WS_NPW;                  // candidates cj and cj + 8
        MG_STAMP_DECL
        const int nql = (D - nroot + 3) >> 2;             // quad lanes per sample
        const int gl = nql + 1;                           // + the root lane
        const int rpi = 64 / gl;                          // samples per wave instruction
        const int fsub = lane / gl, ql = lane - fsub * gl;
        const bool lane_on = lane < rpi * gl;
        const bool root_lane = ql == nql;
        const int d0 = root_lane ? 0 : nroot + 4 * ql;    // first channel of this lane
        const int nst = root_lane ? nroot : (D - d0 < 4 ? D - d0 : 4);
        const int64_t TD = (int64_t)a.T * D;
        const int dp4 = Dp * 4;
        const int lane_img = (d0 + a.cshift) * 4;         // byte offset of the lane's quad inside a basis row
        const int lane_out = fsub * D + d0;               // float offset inside a row group
        // When every quad lane holds four floats and the root lane three (D = 79: 3 + 19 x 4), the root lane borrows the
        // row's channel 3 from quad lane 0 (a cross-lane read) and ALL lanes store four floats with one instruction;
        // the float written twice carries the same value.  Otherwise lanes store 4 / 3 / 2 / 1 floats by class.
        const bool all4 = nroot == 3 && ((D - nroot) & 3) == 0 && !MG_DBG(8192);
        const int q0_lane = (lane - nql) << 2;            // byte index of this row's quad lane 0 for ds_bpermute
        const unsigned lane_out_b = (unsigned)lane_out * 4u;
        int slot = 0;
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            const mg_unit un_prev = mg_unit_at(chunks, a, cur, rot);
            mg_cursor_next(cur, a.n_chunks);
            mg_wait_producers(prog, u + 1);
            MG_STAMP(1);
            if (!MG_DBG(2) && cj < un_prev.ncand) {
                const mg_chunk &ck = un_prev.ck;
                const unsigned char *img = smem + (size_t)slot * buf_bytes;
                const float *lds_ro = (const float *)(ro_base + (size_t)slot * MG_RO_BYTES);
                const float4 *lds_w = (const float4 *)(tb_base + (size_t)slot * MG_TB_BYTES);
                const int *lds_mo = (const int *)(lds_w + max_nt);
                const int col0 = ck.imin * Dp - ck.rt0 * 16;
                const bool has1 = cj + MG_WS_NCW < un_prev.ncand;
                const int c1 = has1 ? cj + MG_WS_NCW : cj;
                const unsigned char *img0 = img + (size_t)(cj * stride + col0) * 4 + lane_img;
                const unsigned char *img1 = img + (size_t)(c1 * stride + col0) * 4 + lane_img;
                const float *ro0 = lds_ro + cj * max_nt * 4, *ro1 = lds_ro + c1 * max_nt * 4;
                float *or0 = out + (size_t)(un_prev.b0 + cj) * TD + (size_t)ck.t0 * D;   // wave-uniform row bases
                float *or1 = out + (size_t)(un_prev.b0 + c1) * TD + (size_t)ck.t0 * D;
                // two row groups x two candidates in flight per trip; the loop exists twice: with the usual row pitch
                // (Dp = 80 floats) as a constant, and with a run-time pitch
                auto sweep_rows = [&](auto pitch_tag) {
                constexpr int DP4 = decltype(pitch_tag)::value;
                for (int f0 = 0; f0 < ck.nT; f0 += 2 * rpi) {
                    const int fla = f0 + fsub, flb = fla + rpi;
                    const bool oa = lane_on && fla < ck.nT, ob = lane_on && flb < ck.nT;
                    const int fa_ = fla < ck.nT ? fla : ck.nT - 1, fb_ = flb < ck.nT ? flb : ck.nT - 1;
                    float *pa0 = or0 + (size_t)f0 * D, *pa1 = or1 + (size_t)f0 * D;          // uniform
                    float *pb0 = pa0 + (size_t)rpi * D, *pb1 = pa1 + (size_t)rpi * D;
                    if (f0 + rpi < ck.nT || MG_DBG(2048)) {   // the usual trip: both row groups (flag 2048: always)
                        f32x4 v0a, v0b, v1a, v1b;
                        if (MG_DBG(4)) {   // ablation: stores only
                            v0a = v0b = v1a = v1b = f32x4{1.f, 2.f, 3.f, 4.f};
                        } else if (!root_lane) {
                            const float4 wa = lds_w[fa_], wb = lds_w[fb_];
                            const int moa = lds_mo[fa_], mob = lds_mo[fb_];
                            if (MG_DBG(131072)) {
                                v0a = mg_quad_taps_t<DP4>(img0 + moa, wa, dp4);
                                v0b = mg_quad_taps_t<DP4>(img0 + mob, wb, dp4);
                                v1a = mg_quad_taps_t<DP4>(img1 + moa, wa, dp4);
                                v1b = mg_quad_taps_t<DP4>(img1 + mob, wb, dp4);
                            } else {   // all 16 tap rows are requested before the first FMA
                                const mg_tap_rows r0a = mg_quad_load<DP4>(img0 + moa, dp4), r0b = mg_quad_load<DP4>(img0 + mob, dp4);
                                const mg_tap_rows r1a = mg_quad_load<DP4>(img1 + moa, dp4), r1b = mg_quad_load<DP4>(img1 + mob, dp4);
                                __builtin_amdgcn_sched_barrier(0);
                                v0a = mg_quad_fma(r0a, wa);
                                v0b = mg_quad_fma(r0b, wb);
                                v1a = mg_quad_fma(r1a, wa);
                                v1b = mg_quad_fma(r1b, wb);
                            }
                        } else {
                            v0a = *(const f32x4 *)&ro0[fa_ * 4];
                            v0b = *(const f32x4 *)&ro0[fb_ * 4];
                            v1a = *(const f32x4 *)&ro1[fa_ * 4];
                            v1b = *(const f32x4 *)&ro1[fb_ * 4];
                        }
                        if (all4 && MG_DBG(4)) {
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (ob) mg_store4_at(pb0, lane_out_b, v0b);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                            if (ob && has1) mg_store4_at(pb1, lane_out_b, v1b);
                        } else if (all4) {
                            const float b0a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0a[0])));
                            const float b0b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0b[0])));
                            const float b1a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1a[0])));
                            const float b1b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1b[0])));
                            if (root_lane) { v0a[3] = b0a; v0b[3] = b0b; v1a[3] = b1a; v1b[3] = b1b; }
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (ob) mg_store4_at(pb0, lane_out_b, v0b);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                            if (ob && has1) mg_store4_at(pb1, lane_out_b, v1b);
                        } else {
                            if (oa) mg_store_n(pa0 + lane_out, v0a, nst);
                            if (ob) mg_store_n(pb0 + lane_out, v0b, nst);
                            if (oa && has1) mg_store_n(pa1 + lane_out, v1a, nst);
                            if (ob && has1) mg_store_n(pb1 + lane_out, v1b, nst);
                        }
                    } else {                                     // the chunk's last rows fill one group only: half the work
                        f32x4 v0a, v1a;
                        if (!root_lane) {
                            const float4 wa = lds_w[fa_];
                            const int moa = lds_mo[fa_];
                            v0a = mg_quad_taps_t<DP4>(img0 + moa, wa, dp4);
                            v1a = mg_quad_taps_t<DP4>(img1 + moa, wa, dp4);
                        } else {
                            v0a = *(const f32x4 *)&ro0[fa_ * 4];
                            v1a = *(const f32x4 *)&ro1[fa_ * 4];
                        }
                        if (all4) {
                            const float b0a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0a[0])));
                            const float b1a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1a[0])));
                            if (root_lane) { v0a[3] = b0a; v1a[3] = b1a; }
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                        } else {
                            if (oa) mg_store_n(pa0 + lane_out, v0a, nst);
                            if (oa && has1) mg_store_n(pa1 + lane_out, v1a, nst);
                        }
                    }
                }
                };
                if (dp4 == 320 && !MG_DBG(16384)) sweep_rows(std::integral_constant<int, 320>{});
                else sweep_rows(std::integral_constant<int, 0>{});
            }
            MG_STAMP(4);
            mg_publish(prog, wave, lane, u + 1);
            if (++slot == nbuf) slot = 0;
            MG_STAMP(5);
        }
        MG_STAMP_DUMP;
    } else if (wave != 0) {
        // ================= f32 producers (waves 1..3): E' fragments -> MFMA -> LDS image =================
        float sfrag[KK];
#pragma unroll
        for (int kk = 0; kk < KK; kk++) sfrag[kk] = 0.f;
        int cur_tile = -1;
        int prev_tile = -1, prev_chunk = -1;
        const float2 *ep = (const float2 *)Epack;
        int slot = 0;
        MG_STAMP_DECL
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            const mg_unit un = mg_unit_at(chunks, a, cur, rot);
            mg_cursor_next(cur, a.n_chunks);
            if (u >= nbuf) mg_wait_consumers(prog, u - nbuf + 1);   // the slot's previous unit has been swept
            MG_STAMP(5);
            if (!MG_DBG(1)) {
                const mg_chunk &ck = un.ck;
                float *lds_c = (float *)(smem + (size_t)slot * buf_bytes);
                if (un.tile != cur_tile) {
                    cur_tile = un.tile;
                    int g_op = g;   // opaque: keeps the (loop-invariant) clamped indices from being hoisted and spilled
                    asm volatile("" : "+v"(g_op));
                    mg_load_sfrag<KK, LAT_F64>(sfrag, lat, un, a.ld, L, cl, g_op);
                }
                // Consecutive chunks of a tile share basis functions (for 'walk' windows of 10 advance by 7): the row tiles this
                // unit has in common with the previous one are copied from the previous slot (LDS -> LDS, the same
                // bits) instead of being recomputed: 37 % fewer MFMAs and E' fragment loads, which is what slows
                // the sweep waves down (matrix-pipe time on the shared SIMDs, L2 requests in the store path).
                int n_ov = 0, src_shift = 0;
                if (un.tile == prev_tile && un.chunk == prev_chunk + 1) {   // the previous unit was this tile's previous chunk
                    const mg_chunk pk = chunks[un.chunk - 1];
                    src_shift = ck.rt0 - pk.rt0;
                    n_ov = pk.rt0 + pk.ntiles - ck.rt0;   // tiles [ck.rt0, pk.rt0 + pk.ntiles) exist in the previous slot
                    n_ov = (n_ov < 0 || src_shift < 0) ? 0 : (n_ov > ck.ntiles ? ck.ntiles : n_ov);   // a grid may run backwards
                }
                if (MG_DBG(4096)) n_ov = 0;   // ablation: no carried-over tiles (every window computed in full)
                // this slot held unit u - nbuf and was the copy source of unit u - nbuf + 1: every row producer must
                // have finished that unit before the slot is overwritten (with two slots: a full meeting per unit)
                if (u >= nbuf - 1) mg_wait_row_producers(prog, u - nbuf + 2);
                mg_produce_f32<KK>(ep, mean32, ck, lds_c, stride, n_ov, wave - 1, MG_WS_NPW - 1, sfrag, lane, cl, g, MG_DBG(256) ? 0 : (int)(blockIdx.x / a.n_chunks),
                                   (MG_DBG(32768) && (u & 1)) ? (a.debug | 512) : a.debug);   // 32768: E' loads on every other unit only
                MG_STAMP(2);
                if (n_ov > 0) {
                    mg_wait_row_producers(prog, u);   // the previous unit's window is complete
                    const float *lds_p = (const float *)(smem + (size_t)(slot == 0 ? nbuf - 1 : slot - 1) * buf_bytes);
                    for (int t = wave - 1; t < n_ov; t += MG_WS_NPW - 1)
                        *(f32x4 *)&lds_c[cl * stride + t * 16 + 4 * g] = *(const f32x4 *)&lds_p[cl * stride + (t + src_shift) * 16 + 4 * g];
                }
                MG_STAMP(3);
            }
            prev_tile = un.tile;
            prev_chunk = un.chunk;
            mg_publish(prog, wave, lane, u + 1);
            if (++slot == nbuf) slot = 0;
            MG_STAMP(4);
        }
        MG_STAMP_DUMP;
    } else {
        // ================= wave 0: tables, root rows and root taps (f64 MFMA), one unit ahead =================
        // per-lane constants of the tap MFMA: B operand = rows[m = 4 ks + (l >> 4)][col = 16 ct + (l & 15)] with
        // col = candidate * nroot + channel (48 columns = 3 tiles); D column = the same col
        int tap_b_off[3][MG_TAP_KS], tap_o_off[3];
        bool tap_b_ok[3][MG_TAP_KS];
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
            const int col = ct * 16 + cl;
            const bool colok = col < MG_NCAND * nroot;
            const int cc = colok ? col / nroot : 0, cd = colok ? col - cc * nroot : 0;
            tap_o_off[ct] = colok ? cc * max_nt * 4 + cd : -1;
#pragma unroll
            for (int ks = 0; ks < MG_TAP_KS; ks++) {
                const int m = 4 * ks + g;
                tap_b_ok[ct][ks] = colok && m < a.max_wi;
                tap_b_off[ct][ks] = cc * root_stride + m * nroot + cd;
            }
        }
        MG_STAMP_DECL
        auto root_stage = [&](const mg_unit &un, int slot) {   // tables -> tb[slot], root rows -> rs, root outputs -> ro[slot]
            const mg_chunk &ck = un.ck;
            float4 *tw = (float4 *)(tb_base + (size_t)slot * MG_TB_BYTES);
            int *tmo = (int *)(tw + max_nt);
            double *rs = (double *)rs_base;
            float4 r_w = {0.f, 0.f, 0.f, 0.f};
            int r_i0 = 0;
            if (lane < ck.nT) { r_w = w32[ck.t0 + lane]; r_i0 = i0tab[ck.t0 + lane]; }
            double r_wt[MG_TAP_FT * MG_TAP_KS];
#pragma unroll
            for (int e = 0; e < MG_TAP_FT * MG_TAP_KS; e++) r_wt[e] = wtap[((size_t)un.chunk * (MG_TAP_FT * MG_TAP_KS) + e) * 64 + lane];
            typename mg_gmm_xt<LAT_F64>::type s64frag[KK];   // widened to float64 at the MFMA
            mg_gmm_load_x<KK, LAT_F64>(s64frag, lat, un.b0, un.ncand, a.ld, L, cl, g);
            if (MG_DBG(32)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MG_STAMP(1); }
            // up to 3 root tiles (8 basis functions x 3 channels = 24 rows, rr = i*nroot + d), chains
            // interleaved; v_mfma_f64_16x16x4_f64 C/D: col = lane & 15, row = (lane >> 4) + 4*reg
            f64x4 racc[3];
            const double *rpp[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int tc = t < ck.nrt ? t : ck.nrt - 1;
                rpp[t] = Erpack + ((size_t)(ck.rrt0 + tc) * KK) * 64 + lane;
                const int row0 = (ck.rrt0 + tc) * 16;
                racc[t][0] = meanroot[row0 + g];
                racc[t][1] = meanroot[row0 + g + 4];
                racc[t][2] = meanroot[row0 + g + 8];
                racc[t][3] = meanroot[row0 + g + 12];
            }
            // all fragments of the 3 tiles in one round of loads (one L2 round trip under store pressure costs
            // thousands of cycles); only for many components in two halves to stay inside the register budget
            constexpr int NH = KK <= 10 ? 1 : 2;
            constexpr int KH = KK / NH;
#pragma unroll
            for (int h = 0; h < NH; h++) {
                double rp[3][KH];
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int q = 0; q < KH; q++) rp[t][q] = rpp[t][(h * KH + q) * 64];
                if (h == 0) {
                    if (lane < ck.nT) {
                        tw[lane] = r_w;
                        tmo[lane] = (r_i0 - ck.imin) * Dp * 4;   // byte offset of the first tap row in the f32 image
                    }
                }
#pragma unroll
                for (int q = 0; q < KH; q++)
#pragma unroll
                    for (int t = 0; t < 3; t++)
                        racc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rp[t][q], (double)s64frag[h * KH + q], racc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int lr0 = (ck.rrt0 + t) * 16 + g - ck.imin * nroot;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int lr = lr0 + 4 * r;
                    if (t < ck.nrt && lr >= 0 && lr < ck.wi * nroot) rs[cl * root_stride + lr] = racc[t][r];
                }
            }
            if (MG_DBG(32)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); MG_STAMP(2); }
            // root taps, again on the float64 matrix pipe: out[f][(c,d)] = sum_m W[f][m] * rows[m][(c,d)] with the
            // banded W[f][m] = w[f][m - m0(f)] (0 outside the 4 taps) pre-packed per chunk as A fragments.  The zero
            // products leave the accumulator untouched and the taps are met in ascending m, so the result is
            // bit-identical to w0*c0, fma(w1,c1,.), fma(w2,c2,.), fma(w3,c3,.).  Same wave wrote rs: program order syncs.
            float *ro = (float *)(ro_base + (size_t)slot * MG_RO_BYTES);
#pragma unroll
            for (int ft = 0; ft < MG_TAP_FT; ft++) {
                if (ft * 16 < ck.nT) {
                    f64x4 acc[3];
                    double bv[3][MG_TAP_KS];
#pragma unroll
                    for (int ct = 0; ct < 3; ct++) {
                        acc[ct] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int ks = 0; ks < MG_TAP_KS; ks++)   // rows at or beyond this chunk's window were never written: 0 * stale LDS could be NaN
                            bv[ct][ks] = (tap_b_ok[ct][ks] && 4 * ks + g < ck.wi) ? rs[tap_b_off[ct][ks]] : 0.0;
                    }
#pragma unroll
                    for (int ks = 0; ks < MG_TAP_KS; ks++)
#pragma unroll
                        for (int ct = 0; ct < 3; ct++)
                            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(r_wt[ft * MG_TAP_KS + ks], bv[ct][ks], acc[ct], 0, 0, 0);
                    // D[row = f = 16 ft + (l >> 4) + 4 reg][col]
#pragma unroll
                    for (int ct = 0; ct < 3; ct++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int fo = ft * 16 + g + 4 * r;
                            if (tap_o_off[ct] >= 0 && fo < ck.nT) ro[tap_o_off[ct] + fo * 4] = (float)acc[ct][r];
                        }
                }
                if (MG_DBG(32) && ft == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); MG_STAMP(6); }
            }
        };
        int slot = 0;
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            const mg_unit un = mg_unit_at(chunks, a, cur, rot);
            mg_cursor_next(cur, a.n_chunks);
            if (u >= nbuf) mg_wait_consumers(prog, u - nbuf + 1);
            MG_STAMP(5);
            if (!MG_DBG(1) && !MG_DBG(1024)) root_stage(un, slot);
            MG_STAMP(3);
            mg_publish(prog, wave, lane, u + 1);
            if (++slot == nbuf) slot = 0;
            MG_STAMP(4);
        }
        MG_STAMP_DUMP;
    }
    if (FUSE_GMM && wave < MG_WS_NPW) {
        // up to four tiles per workgroup, as two groups of two written out one after the other (a loop would let the
        // compiler hoist the exp/log polynomial constants above the MFMA code and spill)
        mg_fused_gmm_terms<KK, LAT_F64>(prog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, wave, lane, 0);
        if (wave < 2) mg_fused_gmm_finish(prog, logp, a.B, a.n_tiles, gK, wave, lane, 0);
        const int64_t my_tiles = ((int64_t)blockIdx.x + 1) * a.n_tiles / gridDim.x - (int64_t)blockIdx.x * a.n_tiles / gridDim.x;
        if (my_tiles > 2) {
            mg_wait_producers(prog + 24, 1);   // gfin[0], gfin[1] (entries 2, 3 are preset): both term buffers are free again
            mg_fused_gmm_terms<KK, LAT_F64>(prog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, wave, lane, 1);
            if (wave < 2) mg_fused_gmm_finish(prog, logp, a.B, a.n_tiles, gK, wave, lane, 1);
        }
    }
}


// -----------------------------------------------------------------------------------------
// The chunk-stationary variant of the hot-path kernel, for batches of many tiles per CU (the default from two
// units per workgroup on; DESIGN.md section 4.1 has the measurements behind every statement here).
//
// The tile-major kernel above ended with its row producers 88 % busy and the sweep waiting for them; 9-13 us of its
// 84-93 went on the E' fragment loads of the unit loop -- 189 MB of L2 hits per launch that travel through the same L2
// and the same per-CU memory pipe as 400 MB of stores (streaming just five more tiles per unit into THIS kernel's row
// producers costs 18 us: MG_DEBUG_FLAGS & 2048).  So E' must not travel at all inside the loop:
//   * a workgroup works on ONE time chunk for its whole life (workgroup w: chunk w mod n_chunks) and walks a block of
//     consecutive candidate tiles; the chunk's whole window of E' fragments is loaded ONCE into registers: 768-thread
//     workgroups, 3 waves per SIMD, 168 VGPRs -- three row producers x TPWP (12) row tiles x KK floats, the four oldest
//     sweep waves TPWS (4) tiles each -- and every unit is MFMAs on registers + LDS writes, no vector-memory read but the
//     2.5 KB of latents (prefetched a unit ahead);
//   * consecutive units are different tiles, so no window rows carry over: each unit computes its full window (51
//     instead of 36 row tiles for 'walk': +42 % MFMAs, cheaper than the loads they replace) and the LDS -> LDS copies
//     disappear;
//   * the per-sample tables, the tap weights and the root rows' float64 fragments are stationary as well (loaded once
//     by wave 0), mean' sits in LDS as the MFMAs' C-in;
//   * wave 0 = root producer, waves 1-3 = row producers, waves 4-11 = sweep, two candidates each exactly as in the
//     tile-major kernel; waves 4-7 -- the older ones, which win the arbiter and finish a unit first -- produce their
//     tiles of the NEXT unit after each sweep (all eight doing so turns the hand-over into a barrier: +4 us); same ring
//     of two LDS slots, same kind of progress counters, same arithmetic: bit-identical results.
// Start-up (the first store leaves ~7 us after kernel entry; 11.4 before the points below): all kernel arguments
// requested in one batch; chunk descriptors in the arguments; no division in the workgroup mapping; every one-time load
// unconditional (clamped index) and scoped to the role that uses it; ONE barrier, which does not drain vector memory,
// between the requests the first root stage waits for and the 130 KB of row fragments; readiness of mean' and of the
// tables handed over through flags; the first root unit peeled from the loop.
// Store order: at any moment the 4 workgroups of a group write the 4 chunks of the same tile, groups are 8 tiles
// apart (stand-alone replica of this order: 64-65 us against 62-64 us for the tile-major order, tools/chan_probe.hip
// mode 3).
// -----------------------------------------------------------------------------------------
#define MG_CS_NPW 4      // producer waves (0: root + latents, 1-3: rows)
#define MG_CS_NCW 8      // sweep waves, two candidates each; they also produce a few row tiles per unit
#ifndef MG_CS_NSP
#define MG_CS_NSP 4      // how many of the sweep waves (the first ones, which sweep fastest) produce row tiles as well
#endif
#define MG_CS_BLOCK (64 * (MG_CS_NPW + MG_CS_NCW))
template <int KK> struct mg_cs_cfg {
    static constexpr int TPWP = 120 / KK < 17 ? 120 / KK : 17;   // row tiles a row producer keeps in registers (TPWP * KK VGPRs)
    static constexpr int TPWS_REGS = 160 / MG_CS_NSP;                            // VGPRs a producing sweep wave spends on fragments
    static constexpr int TPWS = TPWS_REGS / KK < 5 ? (TPWS_REGS / KK > 0 ? TPWS_REGS / KK : 1) : 5;   // row tiles it keeps in registers
    static constexpr int MAX_TILES = (MG_CS_NPW - 1) * TPWP + MG_CS_NSP * TPWS;
};
int mg_cs_max_tiles(int KK) {
    switch (KK) {
        case 2: return mg_cs_cfg<2>::MAX_TILES;   case 4: return mg_cs_cfg<4>::MAX_TILES;   case 6: return mg_cs_cfg<6>::MAX_TILES;
        case 8: return mg_cs_cfg<8>::MAX_TILES;   case 10: return mg_cs_cfg<10>::MAX_TILES; case 12: return mg_cs_cfg<12>::MAX_TILES;
        case 14: return mg_cs_cfg<14>::MAX_TILES; case 16: return mg_cs_cfg<16>::MAX_TILES; default: return 0;
    }
}

// progress counters of the chunk-stationary kernel (LDS ints): [0..11] units produced by wave w (every wave produces
// row tiles; wave 0 the root rows), [12] latent tiles staged by wave 0, [16..23] units swept by sweep wave 4 + i;
// the mixture's hand-off uses [32..63]
#define MG_CS_PROG_LAT 12
#define MG_CS_PROG_SWEPT 16
#define MG_CS_PROG_MEAN 24   // [24..27]: mean' of the window's rows copied by sweep wave 4 + MG_CS_NSP + i
#define MG_CS_PROG_GMM 32
#define MG_CS_PROG_INTS 64
__device__ __forceinline__ void mg_cs_wait_produced(const mg_lds_int *prog, int target) {   // the producing waves: 0 .. 3 + MG_CS_NSP
    for (;;) {
        const i32x4 v = *(const volatile mg_lds_i32x4 *)prog;
        const i32x4 x = *(const volatile mg_lds_i32x4 *)(prog + 4);
        int m = min(min(min(v[0], v[1]), min(v[2], v[3])), min(min(x[0], x[1]), min(x[2], x[3])));
        if (MG_CS_NSP > 4) {
            const i32x4 y = *(const volatile mg_lds_i32x4 *)(prog + 8);
            m = min(m, min(min(y[0], y[1]), min(y[2], y[3])));
        }
        if (__builtin_amdgcn_readfirstlane(m) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void mg_cs_wait_swept(const mg_lds_int *prog, int target) {   // the eight sweep waves
    for (;;) {
        const i32x4 v = *(const volatile mg_lds_i32x4 *)(prog + MG_CS_PROG_SWEPT);
        const i32x4 x = *(const volatile mg_lds_i32x4 *)(prog + MG_CS_PROG_SWEPT + 4);
        const int m = min(min(min(v[0], v[1]), min(v[2], v[3])), min(min(x[0], x[1]), min(x[2], x[3])));
        if (__builtin_amdgcn_readfirstlane(m) >= target) break;
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void mg_cs_wait_mean(const mg_lds_int *prog) {   // the (at most four) copying sweep waves
    for (;;) {
        const i32x4 v = *(const volatile mg_lds_i32x4 *)(prog + MG_CS_PROG_MEAN);
        if (__builtin_amdgcn_readfirstlane(min(min(v[0], v[1]), min(v[2], v[3]))) >= 1) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void mg_cs_wait_latents(const mg_lds_int *prog, int target) {
    for (;;) {
        const int v = *(const volatile mg_lds_int *)(prog + MG_CS_PROG_LAT);
        if (__builtin_amdgcn_readfirstlane(v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// NT row tiles t = first + step * i (i < NT, t < ntiles) of a unit: D = E' tile . latent tile + mean', the E' fragments in
// registers, the latent tile's B fragments and mean' (C-in) from LDS; chains of two tiles interleaved
template <int KK, int NT>
__device__ __forceinline__ void mg_cs_produce(const float (&ef)[NT][KK], const float *lds_lat, const float *lds_mean, float *lds_c,
                                              int stride, int first, int step, int ntiles, int lane, int cl, int g) {
    if (first >= ntiles) return;
    float sfrag[KK];
#pragma unroll
    for (int kk = 0; kk < KK; kk++) sfrag[kk] = lds_lat[kk * 64 + lane];
#pragma unroll
    for (int i = 0; i < NT; i += 2) {
        const int t0 = first + step * i, t1 = t0 + step;
        if (t0 < ntiles) {
            const int i1 = (i + 1 < NT) ? i + 1 : i;
            const bool on1 = (i + 1 < NT) && t1 < ntiles;
            f32x4 acc0 = *(const f32x4 *)&lds_mean[t0 * 16 + 4 * g];
            if (on1) {
                f32x4 acc1 = *(const f32x4 *)&lds_mean[t1 * 16 + 4 * g];
#pragma unroll
                for (int kk = 0; kk < KK; kk++) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[i][kk], sfrag[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[i1][kk], sfrag[kk], acc1, 0, 0, 0);
                }
                *(f32x4 *)&lds_c[cl * stride + t0 * 16 + 4 * g] = acc0;   // D[row = 4g + reg][col = cl]: four consecutive padded rows
                *(f32x4 *)&lds_c[cl * stride + t1 * 16 + 4 * g] = acc1;
            } else {
#pragma unroll
                for (int kk = 0; kk < KK; kk++) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[i][kk], sfrag[kk], acc0, 0, 0, 0);
                *(f32x4 *)&lds_c[cl * stride + t0 * 16 + 4 * g] = acc0;
            }
        }
    }
}
template <int KK, int NT>
__device__ __forceinline__ void mg_cs_load_fragments(float (&ef)[NT][KK], const float2 *ep, const mg_chunk &ck, int first, int step, int lane) {
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int t = first + step * i;
        const int tc = t < ck.ntiles ? t : ck.ntiles - 1;   // clamp: redundant but in bounds
        const float2 *p = ep + ((size_t)(ck.rt0 + tc) * (KK / 2)) * 64 + lane;
#pragma unroll
        for (int q2 = 0; q2 < KK / 2; q2++) {
            const float2 v = p[q2 * 64];
            ef[i][2 * q2] = v.x;
            ef[i][2 * q2 + 1] = v.y;
        }
    }
}

template <int KK, bool LAT_F64, bool FUSE_GMM>
__global__ __launch_bounds__(MG_CS_BLOCK) void mg_frames_cs_kernel(
    const float *__restrict__ Epack,      // [RT][KK/2][64][2]
    const float *__restrict__ mean32,     // [RT*16]
    const double *__restrict__ Erpack,    // [RRT][KK][64]
    const double *__restrict__ meanroot,  // [RRT*16]
    const void *__restrict__ lat,         // (B, ld) f32 or f64
    const int32_t *__restrict__ i0tab,    // (T)
    const float4 *__restrict__ w32,       // (T)
    const double *__restrict__ wtap,      // [n_chunks][FT][KS][64]
    const mg_chunk *__restrict__ chunks,
    float *__restrict__ out,              // (B,T,D)
    const double *__restrict__ gPpack, const double *__restrict__ gmP, const double *__restrict__ gcst,
    float *__restrict__ logp,             // FUSE_GMM: (B) log p(s_b)
    const mg_frames_args a, const int gK, const int gJT, const int buf_bytes) {
    constexpr int TPWP = mg_cs_cfg<KK>::TPWP, TPWS = mg_cs_cfg<KK>::TPWS;
    constexpr int NRP = MG_CS_NPW - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave == 0) MG_SUB_STAMP(14, 0, 0);
    // Kernel arguments: left alone, the compiler fetches each where a role first uses it -- a dependent trip to memory per
    // 64-byte line of the argument block (wave 0 alone made eight in a row before it had issued its loads, 4.8 us after entry).
    // Asking for all of them here makes that one trip; the later uses hit the scalar cache.
    asm volatile("" ::"s"(Epack), "s"(mean32), "s"(Erpack), "s"(meanroot), "s"(lat), "s"(i0tab), "s"(w32), "s"(wtap), "s"(out), "s"(gPpack),
                 "s"(gmP), "s"(gcst), "s"(logp), "s"(a.B), "s"(a.ld), "s"(a.T), "s"(a.L), "s"(a.n_chunks), "s"(a.stride), "s"(a.max_tiles),
                 "s"(a.ck[1].t0), "s"(a.ck[3].t0), "s"(a.ck[5].t0), "s"(a.ck[7].t0), "s"(gK), "s"(gJT), "s"(buf_bytes));
    const int stride = a.stride, D = a.D, Dp = a.Dp, L = a.L, nroot = a.nroot;
    const int root_stride = a.max_wi * nroot + 1;
    const int max_nt = a.max_nt;
    const int RO_BYTES = MG_RO_BYTES_N(max_nt), TB_BYTES = MG_TB_BYTES_N(max_nt);
    unsigned char *ro_base = smem + 2 * (size_t)buf_bytes;             // root outputs, one per ring slot
    unsigned char *tb_base = ro_base + 2 * RO_BYTES;                   // per-sample tables of THE chunk
    unsigned char *rs_base = tb_base + TB_BYTES;                       // float64 root image (wave 0 only)
    float *lds_mean = (float *)(rs_base + (size_t)MG_NCAND * root_stride * 8);   // mean' of the window's rows [max_tiles * 16]
    double *lds_rwt = (double *)((unsigned char *)lds_mean + (size_t)a.max_tiles * 64);   // the chunk's banded tap weights [FT * KS][64]
    double *lds_rmean = lds_rwt + MG_TAP_FT * MG_TAP_KS * 64;                            // mean of the root rows as the root chains' C-in [3][4][4] (wave 0)
    float *lds_latb = (float *)(lds_rmean + 64);                                         // latent tiles as MFMA B fragments [2][KK][64]
    mg_lds_int *prog = (mg_lds_int *)(lds_latb + 2 * KK * 64);                           // MG_CS_PROG_INTS counters, then the mixture's buffers
    if (tid < MG_CS_PROG_INTS)   // never-waited-for entries: the sweep waves that produce nothing, the padding of the mixture's hand-off
        prog[tid] = ((tid >= MG_CS_NPW + MG_CS_NSP && tid < MG_CS_NPW + MG_CS_NCW) || (tid >= MG_CS_PROG_MEAN + (MG_CS_NCW - MG_CS_NSP) && tid < MG_CS_PROG_MEAN + 4) ||
                     tid == MG_CS_PROG_GMM + 26 || tid == MG_CS_PROG_GMM + 27) ? 0x7fffffff : 0;

    // workgroup w: chunk w mod n_chunks; the gridDim / n_chunks workgroups of a chunk share the tiles out in consecutive blocks
    // (the grid is a multiple of n_chunks; the quotient by multiplication, exact for w * n_chunks < 2^20)
    const int n_chunks = a.n_chunks;
    const int q = (int)(((unsigned)blockIdx.x * (unsigned)a.cs_magic) >> 20), c = (int)blockIdx.x - q * n_chunks;
    const int t_begin = q * a.cs_per + (q < a.cs_rem ? q : a.cs_rem);   // the first cs_rem workgroups of a chunk take one tile more
    const int n_units = a.cs_per + (q < a.cs_rem ? 1 : 0);
    const mg_chunk ck = a.ck[c];   // n_chunks <= MG_ARG_CHUNKS where this kernel is launched
    const int cl = lane & 15, g = lane >> 4;
    const int nt_p = ck.ntiles < NRP * TPWP ? ck.ntiles : NRP * TPWP;   // tiles [0, nt_p): row producers; [nt_p, ntiles): sweep waves
    const float2 *ep = (const float2 *)Epack;
    // Start-up.  The CU's one path to memory serves requests in the order they were issued, so first goes what the first unit's
    // root stage waits for (wave 0: latents, root fragments, tables -- 63 requests) and mean' of the window's rows (the sweep
    // waves that produce nothing), THEN the 130 KB of row fragments: one barrier, which does not drain vector memory, separates
    // the two and publishes the zeroed counters.  Everything after it is data flow through those counters: the tables are in
    // LDS before wave 0 publishes its first unit, mean' before its copiers raise their flags, and the row producers start their
    // first MFMAs as the first fragments land while the root stage of the first unit is already running.
    static_assert(MG_CS_NSP < MG_CS_NCW && MG_CS_NCW - MG_CS_NSP <= 4, "the (at most four) sweep waves that produce nothing copy mean'");
    if (wave >= MG_CS_NPW) {
        // ================= sweep waves: two candidates each; the four oldest also produce TPWS row tiles of the NEXT unit =================
        const int cj = wave - MG_CS_NPW;                  // candidates cj and cj + 8
        // the store stream goes before the producers' and the mixture's MFMA chains wherever both are ready (-1 %: 79.4 against
        // 80.2 us; the four younger sweep waves above the four older, producing ones: +3 us)
        __builtin_amdgcn_s_setprio(3);
        float ef[TPWS][KK];
        const bool producing = cj < MG_CS_NSP;
        // the waves that produce nothing copy mean' of the window's rows: unconditional loads at clamped indices (a predicated load
        // drags a wait for everything in flight behind it), all in flight before the first LDS write
        constexpr int MEAN_NTH = 64 * (MG_CS_NCW - MG_CS_NSP);
        const int n_mean = ck.ntiles * 16, e0 = tid - 64 * (MG_CS_NPW + MG_CS_NSP);
        float mv[6];
        if (!producing) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int e = e0 + i * MEAN_NTH;
                mv[i] = mean32[(size_t)ck.rt0 * 16 + (e < n_mean ? e : n_mean - 1)];
            }
        }
        mg_lds_barrier();
        if (producing) {
            mg_cs_load_fragments<KK, TPWS>(ef, ep, ck, nt_p + cj, MG_CS_NSP, lane);   // in flight across barrier B
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (e0 + i * MEAN_NTH < n_mean) lds_mean[e0 + i * MEAN_NTH] = mv[i];
            for (int e = e0 + 6 * MEAN_NTH; e < n_mean; e += MEAN_NTH) lds_mean[e] = mean32[(size_t)ck.rt0 * 16 + e];   // (windows beyond 6 * MEAN_NTH rows)
            mg_publish(prog + MG_CS_PROG_MEAN, cj - MG_CS_NSP, lane, 1);
        }
        const int nql = (D - nroot + 3) >> 2;             // quad lanes per sample
        const int gl = nql + 1;                           // + the root lane
        const int rpi = 64 / gl;                          // samples per wave instruction
        const int fsub = lane / gl, ql = lane - fsub * gl;
        const bool lane_on = lane < rpi * gl;
        const bool root_lane = ql == nql;
        const int d0 = root_lane ? 0 : nroot + 4 * ql;    // first channel of this lane
        const int nst = root_lane ? nroot : (D - d0 < 4 ? D - d0 : 4);
        const int64_t TD = (int64_t)a.T * D;
        const int dp4 = Dp * 4;
        const int lane_img = (d0 + a.cshift) * 4;         // byte offset of the lane's quad inside a basis row
        const int lane_out = fsub * D + d0;               // float offset inside a row group
        const bool all4 = nroot == 3 && ((D - nroot) & 3) == 0;   // every lane stores four floats (see the tile-major kernel)
        const int q0_lane = (lane - nql) << 2;            // byte index of this row's quad lane 0 for ds_bpermute
        const unsigned lane_out_b = (unsigned)lane_out * 4u;
        const float4 *lds_w = (const float4 *)tb_base;
        const int *lds_mo = (const int *)(lds_w + max_nt);
        const int col0 = ck.imin * Dp - ck.rt0 * 16;
        if (wave == 8) MG_SUB_STAMP(15, 2, 0);
        if (wave == 4) MG_SUB_STAMP(15, 2, 1);
        MG_STAMP_DECL
        if (n_units > 0 && producing) {   // this wave's tiles of the first unit
            mg_cs_wait_mean(prog);
            mg_cs_wait_latents(prog, 1);
            mg_cs_produce<KK, TPWS>(ef, lds_latb, lds_mean, (float *)smem, stride, nt_p + cj, MG_CS_NSP, ck.ntiles, lane, cl, g);
            mg_publish(prog, wave, lane, 1);
        }
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            const int64_t b0 = (int64_t)(t_begin + u) * MG_NCAND;
            const int ncand = (int)((a.B - b0) < MG_NCAND ? (a.B - b0) : MG_NCAND);
            mg_cs_wait_produced(prog, u + 1);
            MG_STAMP(1);
            MG_UNIT_STAMP(u, 0);
            const int slot = u & 1;
            const unsigned char *img = smem + (size_t)slot * buf_bytes;
            const float *lds_ro = (const float *)(ro_base + (size_t)slot * RO_BYTES);
            const bool has1 = cj + 8 < ncand;
            const int c1 = has1 ? cj + 8 : cj;
            const unsigned char *img0 = img + (size_t)(cj * stride + col0) * 4 + lane_img;
            const unsigned char *img1 = img + (size_t)(c1 * stride + col0) * 4 + lane_img;
            const float *ro0 = lds_ro + cj * max_nt * 4, *ro1 = lds_ro + c1 * max_nt * 4;
            float *or0 = out + (size_t)(b0 + cj) * TD + (size_t)ck.t0 * D;   // wave-uniform row bases
            float *or1 = out + (size_t)(b0 + c1) * TD + (size_t)ck.t0 * D;
            auto sweep_rows = [&](auto pitch_tag, int f_first, int f_last) {
                constexpr int DP4 = decltype(pitch_tag)::value;
                for (int f0 = f_first; f0 < f_last; f0 += 2 * rpi) {
                    const int fla = f0 + fsub, flb = fla + rpi;
                    const bool oa = lane_on && fla < ck.nT, ob = lane_on && flb < ck.nT;
                    const int fa_ = fla < ck.nT ? fla : ck.nT - 1, fb_ = flb < ck.nT ? flb : ck.nT - 1;
                    float *pa0 = or0 + (size_t)f0 * D, *pa1 = or1 + (size_t)f0 * D;          // uniform
                    float *pb0 = pa0 + (size_t)rpi * D, *pb1 = pa1 + (size_t)rpi * D;
                    if (f0 + rpi < ck.nT) {   // the usual trip: both row groups
                        f32x4 v0a, v0b, v1a, v1b;
                        if (MG_DBG(4)) {   // ablation: stores only
                            v0a = v0b = v1a = v1b = f32x4{1.f, 2.f, 3.f, 4.f};
                        } else if (!root_lane) {   // all 16 tap rows are requested before the first FMA
                            const float4 wa = lds_w[fa_], wb = lds_w[fb_];
                            const int moa = lds_mo[fa_], mob = lds_mo[fb_];
                            const mg_tap_rows r0a = mg_quad_load<DP4>(img0 + moa, dp4), r0b = mg_quad_load<DP4>(img0 + mob, dp4);
                            const mg_tap_rows r1a = mg_quad_load<DP4>(img1 + moa, dp4), r1b = mg_quad_load<DP4>(img1 + mob, dp4);
                            __builtin_amdgcn_sched_barrier(0);
                            v0a = mg_quad_fma(r0a, wa);
                            v0b = mg_quad_fma(r0b, wb);
                            v1a = mg_quad_fma(r1a, wa);
                            v1b = mg_quad_fma(r1b, wb);
                        } else {
                            v0a = *(const f32x4 *)&ro0[fa_ * 4];
                            v0b = *(const f32x4 *)&ro0[fb_ * 4];
                            v1a = *(const f32x4 *)&ro1[fa_ * 4];
                            v1b = *(const f32x4 *)&ro1[fb_ * 4];
                        }
                        if (all4 && MG_DBG(4)) {
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (ob) mg_store4_at(pb0, lane_out_b, v0b);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                            if (ob && has1) mg_store4_at(pb1, lane_out_b, v1b);
                        } else if (all4) {
                            const float b0a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0a[0])));
                            const float b0b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0b[0])));
                            const float b1a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1a[0])));
                            const float b1b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1b[0])));
                            if (root_lane) { v0a[3] = b0a; v0b[3] = b0b; v1a[3] = b1a; v1b[3] = b1b; }
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (ob) mg_store4_at(pb0, lane_out_b, v0b);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                            if (ob && has1) mg_store4_at(pb1, lane_out_b, v1b);
                        } else {
                            if (oa) mg_store_n(pa0 + lane_out, v0a, nst);
                            if (ob) mg_store_n(pb0 + lane_out, v0b, nst);
                            if (oa && has1) mg_store_n(pa1 + lane_out, v1a, nst);
                            if (ob && has1) mg_store_n(pb1 + lane_out, v1b, nst);
                        }
                    } else {                                     // the chunk's last rows fill one group only: half the work
                        f32x4 v0a, v1a;
                        if (!root_lane) {
                            const float4 wa = lds_w[fa_];
                            const int moa = lds_mo[fa_];
                            v0a = mg_quad_taps_t<DP4>(img0 + moa, wa, dp4);
                            v1a = mg_quad_taps_t<DP4>(img1 + moa, wa, dp4);
                        } else {
                            v0a = *(const f32x4 *)&ro0[fa_ * 4];
                            v1a = *(const f32x4 *)&ro1[fa_ * 4];
                        }
                        if (all4) {
                            const float b0a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0a[0])));
                            const float b1a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1a[0])));
                            if (root_lane) { v0a[3] = b0a; v1a[3] = b1a; }
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                        } else {
                            if (oa) mg_store_n(pa0 + lane_out, v0a, nst);
                            if (oa && has1) mg_store_n(pa1 + lane_out, v1a, nst);
                        }
                    }
                }
            };
            const bool mine = cj < ncand && !MG_DBG(2);
            if (mine) {
                if (dp4 == 320) sweep_rows(std::integral_constant<int, 320>{}, 0, ck.nT);
                else sweep_rows(std::integral_constant<int, 0>{}, 0, ck.nT);
            }
            MG_STAMP(4);
            MG_UNIT_STAMP(u, 1);
            mg_publish(prog + MG_CS_PROG_SWEPT, cj, lane, u + 1);
            MG_STAMP(5);
            if (u + 1 < n_units && producing) {
                // the producing sweep waves are the four oldest, which finish a unit first: its row tiles of the NEXT unit go
                // into the slot the unit before this one was swept from (every sweep wave is past it by now, as a rule)
                mg_cs_wait_swept(prog, u);
                mg_cs_wait_latents(prog, u + 2);
                MG_STAMP(3);
                mg_cs_produce<KK, TPWS>(ef, lds_latb + ((u + 1) & 1) * KK * 64, lds_mean, (float *)(smem + (size_t)((u + 1) & 1) * buf_bytes), stride,
                                        nt_p + cj, MG_CS_NSP, ck.ntiles, lane, cl, g);
                mg_publish(prog, wave, lane, u + 2);
                MG_STAMP(2);
            }
        }
        MG_STAMP_DUMP;
    } else if (wave != 0) {
        // ================= row producers (waves 1..3): TPWP row tiles each, the fragments in registers =================
        const int pw = wave - 1;
        float ef[TPWP][KK];
        mg_lds_barrier();
        MG_SUB_STAMP(14 + (wave == 1 ? 1 : 0), wave == 1 ? 0 : 1, 0);
        mg_cs_load_fragments<KK, TPWP>(ef, ep, ck, pw, NRP, lane);   // in tile order: the first unit's MFMAs start as its first fragments land
        if (MG_DBG(32)) {   // when do the fragments land?
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MG_SUB_STAMP(14 + (wave == 1 ? 1 : 0), wave == 1 ? 0 : 1, 1);
        }
        mg_cs_wait_mean(prog);
        MG_STAMP_DECL
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            if (u >= 2) mg_cs_wait_swept(prog, u - 1);   // the slot's previous unit has been swept
            mg_cs_wait_latents(prog, u + 1);
            MG_STAMP(5);
            MG_UNIT_STAMP(u, 0);
            if (!MG_DBG(1))
                mg_cs_produce<KK, TPWP>(ef, lds_latb + (u & 1) * KK * 64, lds_mean, (float *)(smem + (size_t)(u & 1) * buf_bytes), stride, pw, NRP, nt_p,
                                        lane, cl, g);
            MG_STAMP(2);
            MG_UNIT_STAMP(u, 1);
            mg_publish(prog, wave, lane, u + 1);
            MG_STAMP(4);
            if (MG_DBG(2048)) {   // experiment: what would streaming five more row tiles' fragments per unit from L2 cost?
                for (int i = 0; i < 5; i++) {
                    const int tc = min(nt_p + pw * 5 + i, ck.ntiles - 1);
                    const float2 *pp = ep + ((size_t)(ck.rt0 + tc) * (KK / 2)) * 64 + lane;
#pragma unroll
                    for (int q2 = 0; q2 < KK / 2; q2++) { const float2 v = pp[q2 * 64]; asm volatile("" ::"v"(v.x), "v"(v.y)); }
                }
            }
        }
        MG_STAMP_DUMP;
    } else {
        // ================= wave 0: the chunk's tables (once); per unit the latent tile -> LDS, root rows and root taps (f64 MFMA) =================
        // every load first, the ones the first unit's root chains wait for at the head of the queue; the index arithmetic after them
        auto load_latents = [&](typename mg_gmm_xt<LAT_F64>::type (&x)[KK], int u) {
            int t = t_begin + (u < n_units ? u : n_units - 1);    // clamped: the prefetch of the unit after the last one,
            t = t < 0 ? 0 : (t >= a.n_tiles ? a.n_tiles - 1 : t);  // a workgroup without units
            const int64_t b0 = (int64_t)t * MG_NCAND;
            const int ncand = (int)((a.B - b0) < MG_NCAND ? (a.B - b0) : MG_NCAND);
            mg_gmm_load_x<KK, LAT_F64>(x, lat, b0, ncand, a.ld, L, cl, g);
        };
        typename mg_gmm_xt<LAT_F64>::type s64frag[KK], s64next[KK];
        load_latents(s64next, 0);
        asm volatile("" ::: "memory");   // keep this issue order: results return in it
        const double *rpp[3];
        double rm_v[3][4];   // mean of the root rows: the C-in of the root chains, parked in LDS
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int tc = t < ck.nrt ? t : ck.nrt - 1;
            rpp[t] = Erpack + ((size_t)(ck.rrt0 + tc) * KK) * 64 + lane;
        }
        // the root rows' float64 fragments stay in registers as well where the budget allows (6 KK VGPRs)
        constexpr bool RR = KK <= 10;
        double rp_reg[3][RR ? KK : 1];
        if (RR) {
#pragma unroll
            for (int q2 = 0; q2 < KK; q2++)
#pragma unroll
                for (int t = 0; t < 3; t++) rp_reg[t][RR ? q2 : 0] = rpp[t][q2 * 64];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int tc = t < ck.nrt ? t : ck.nrt - 1;
            const double *rmp = meanroot + (ck.rrt0 + tc) * 16 + g;
#pragma unroll
            for (int r = 0; r < 4; r++) rm_v[t][r] = rmp[4 * r];
        }
        double wt_v[MG_TAP_FT * MG_TAP_KS];
#pragma unroll
        for (int e = 0; e < MG_TAP_FT * MG_TAP_KS; e++) wt_v[e] = wtap[((size_t)c * (MG_TAP_FT * MG_TAP_KS) + e) * 64 + lane];
        const int tl = lane < ck.nT ? lane : ck.nT - 1;   // clamped, unconditional (see mv above)
        const float4 tw_v = w32[ck.t0 + tl];
        const int ti_v = i0tab[ck.t0 + tl];
        mg_lds_barrier();
        MG_SUB_STAMP(15, 1, 0);
        // LDS offsets of the root stage, every access unconditional: what must not count reads a zero (lds_rmean[63]), what must
        // not land goes to a spare slot (the padding double of a candidate's root image row; the fourth float of a root output)
        double *rs = (double *)rs_base;
        const int rs_zero = (int)((lds_rmean + 63) - rs);
        int tap_b_off[3][MG_TAP_KS], tap_o_off[3], rs_off[3][4];
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
            const int col = ct * 16 + cl;
            const bool colok = col < MG_NCAND * nroot;
            const int cc = colok ? col / nroot : 0, cd = colok ? col - cc * nroot : 0;
            tap_o_off[ct] = colok ? cc * max_nt * 4 + cd : 3;
#pragma unroll
            for (int ks = 0; ks < MG_TAP_KS; ks++) {
                const int m = 4 * ks + g;   // rows at or beyond the window are never written: 0 * stale LDS could be NaN
                tap_b_off[ct][ks] = (colok && m < ck.wi) ? cc * root_stride + m * nroot + cd : rs_zero;
            }
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int lr0 = (ck.rrt0 + t) * 16 + g - ck.imin * nroot;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int lr = lr0 + 4 * r;
                rs_off[t][r] = cl * root_stride + ((t < ck.nrt && lr >= 0 && lr < ck.wi * nroot) ? lr : a.max_wi * nroot);
            }
        }
        auto park_tables = [&]() {   // the loads issued last, written when the first unit's root chains no longer wait behind them
            float4 *tw = (float4 *)tb_base;
            int *tmo = (int *)(tw + max_nt);
#pragma unroll
            for (int e = 0; e < MG_TAP_FT * MG_TAP_KS; e++) lds_rwt[e * 64 + lane] = wt_v[e];
            if (lane < ck.nT) {
                tw[lane] = tw_v;
                tmo[lane] = (ti_v - ck.imin) * Dp * 4;   // byte offset of the first tap row in the f32 image
            }
        };
        if (lane == 63) lds_rmean[63] = 0.0;
        MG_SUB_STAMP(15, 1, 1);
        MG_SUB_STAMP(14, 0, 1);
        MG_STAMP_DECL
        // one unit of the root stage; the first one is peeled (FIRST) so that what only it uses -- the one-time loads still in
        // registers -- is dead in the loop over the others
        auto root_unit = [&](const int u, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            MG_STAMP(0);
#pragma unroll
            for (int kk = 0; kk < KK; kk++) s64frag[kk] = s64next[kk];
            if (u >= 2) mg_cs_wait_swept(prog, u - 1);   // every wave is done with the slot's previous unit: image, root outputs, latent tile
            MG_STAMP(5);
            MG_UNIT_STAMP(u, 0);
            {   // the latent tile as float32 MFMA B fragments for all the other waves
                float *lb = lds_latb + (u & 1) * KK * 64;
#pragma unroll
                for (int kk = 0; kk < KK; kk++) lb[kk * 64 + lane] = (float)s64frag[kk];
                mg_publish(prog, MG_CS_PROG_LAT, lane, u + 1);
            }
            // the next unit's latents, a unit ahead -- but not yet in the first unit: the request would queue behind the row
            // fragments still being issued and hold this wave up; there it goes out after the root stage
            if (!FIRST) load_latents(s64next, u + 1);
            MG_SUB_STAMP(12, u, 0);
            if (MG_DBG(1024)) {   // ablation: no root stage
                if (FIRST) { park_tables(); load_latents(s64next, 1); }
                MG_STAMP(3); mg_publish(prog, wave, lane, u + 1); MG_STAMP(4);
                return;
            }
            // up to 3 root tiles (rows rr = i*nroot + d), chains interleaved; v_mfma_f64_16x16x4_f64 C/D: col = lane & 15, row = (lane >> 4) + 4*reg
            f64x4 racc[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                if (FIRST) {   // straight from the registers they were loaded into; parked in LDS for the later units
                    racc[t] = f64x4{rm_v[t][0], rm_v[t][1], rm_v[t][2], rm_v[t][3]};
                    if (cl == 0) {   // one lane per row group g: [t][r][g]
#pragma unroll
                        for (int r = 0; r < 4; r++) lds_rmean[(t * 4 + r) * 4 + g] = rm_v[t][r];
                    }
                } else {
                    racc[t] = f64x4{lds_rmean[(t * 4 + 0) * 4 + g], lds_rmean[(t * 4 + 1) * 4 + g], lds_rmean[(t * 4 + 2) * 4 + g], lds_rmean[(t * 4 + 3) * 4 + g]};
                }
            }
            if (RR) {
#pragma unroll
                for (int q2 = 0; q2 < KK; q2++)
#pragma unroll
                    for (int t = 0; t < 3; t++)
                        racc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rp_reg[t][RR ? q2 : 0], (double)s64frag[q2], racc[t], 0, 0, 0);
            } else {
                constexpr int KH = KK / 2;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    double rp[3][KH];
#pragma unroll
                    for (int t = 0; t < 3; t++)
#pragma unroll
                        for (int q2 = 0; q2 < KH; q2++) rp[t][q2] = rpp[t][(h * KH + q2) * 64];
#pragma unroll
                    for (int q2 = 0; q2 < KH; q2++)
#pragma unroll
                        for (int t = 0; t < 3; t++)
                            racc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rp[t][q2], (double)s64frag[h * KH + q2], racc[t], 0, 0, 0);
                }
            }
            MG_SUB_STAMP(12, u, 1);
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) rs[rs_off[t][r]] = racc[t][r];
            if (FIRST) park_tables();
            // root taps on the float64 matrix pipe against the chunk's banded weight matrix (see the tile-major kernel)
            MG_SUB_STAMP(13, u, 0);
            float *ro = (float *)(ro_base + (size_t)(u & 1) * RO_BYTES);
            double bv[3][MG_TAP_KS];
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
#pragma unroll
                for (int ks = 0; ks < MG_TAP_KS; ks++) bv[ct][ks] = rs[tap_b_off[ct][ks]];
#pragma unroll
            for (int ft = 0; ft < MG_TAP_FT; ft++) {
                if (ft * 16 < ck.nT) {   // (rows of the last tile beyond the chunk land in the root outputs' padding: max_nt is a multiple of 16)
                    f64x4 acc[3];
#pragma unroll
                    for (int ct = 0; ct < 3; ct++) acc[ct] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < MG_TAP_KS; ks++)
#pragma unroll
                        for (int ct = 0; ct < 3; ct++)
                            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(lds_rwt[(ft * MG_TAP_KS + ks) * 64 + lane], bv[ct][ks], acc[ct], 0, 0, 0);
#pragma unroll
                    for (int ct = 0; ct < 3; ct++)
#pragma unroll
                        for (int r = 0; r < 4; r++) ro[tap_o_off[ct] + (ft * 16 + g + 4 * r) * 4] = (float)acc[ct][r];
                }
            }
            MG_STAMP(3);
            MG_UNIT_STAMP(u, 1);
            MG_SUB_STAMP(13, u, 1);
            mg_publish(prog, wave, lane, u + 1);
            if (FIRST) load_latents(s64next, 1);
            MG_STAMP(4);
        };
        if (n_units > 0) root_unit(0, std::true_type{});
        for (int u = 1; u < n_units; u++) root_unit(u, std::false_type{});
        MG_STAMP_DUMP;
    }
    if (FUSE_GMM && wave < MG_WS_NPW) {   // the mixture: the four producer waves, as in the tile-major kernel
        mg_lds_int *gprog = prog + MG_CS_PROG_GMM;
        mg_fused_gmm_terms<KK, LAT_F64>(gprog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, wave, lane, 0);
        if (wave < 2) mg_fused_gmm_finish(gprog, logp, a.B, a.n_tiles, gK, wave, lane, 0);
        const int64_t my_tiles = ((int64_t)blockIdx.x + 1) * a.n_tiles / gridDim.x - (int64_t)blockIdx.x * a.n_tiles / gridDim.x;
        if (my_tiles > 2) {
            mg_wait_producers(gprog + 24, 1);   // gfin[0], gfin[1]: both term buffers are free again
            mg_fused_gmm_terms<KK, LAT_F64>(gprog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, wave, lane, 1);
            if (wave < 2) mg_fused_gmm_finish(gprog, logp, a.B, a.n_tiles, gK, wave, lane, 1);
        }
    }
}

// -----------------------------------------------------------------------------------------
// Direct kernel: one thread per output element (b, f, d).  Used for small batches, shapes the
// LDS-staged kernel does not cover, and the all-float64 variant behind the single-sample
// adaptor calls.  Same arithmetic contract as the MFMA kernel (bit-identical float32 results).
// -----------------------------------------------------------------------------------------
struct mg_direct_args {
    const float *Et32;    // [L][R]
    const double *Et64;   // [L][R]
    const double *mean;   // (R)
    const void *lat;
    const int32_t *i0;
    const double *w;
    void *out;
    int64_t B, ld;
    int32_t T, D, L, R, nroot;
};

template <bool LAT_F64, bool OUT_F64>
__global__ __launch_bounds__(256) void mg_frames_direct_kernel(mg_direct_args a) {
    const int64_t TD = (int64_t)a.T * a.D;
    const int64_t total = a.B * TD;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = idx / TD;
        const int o = (int)(idx - b * TD);
        const int f = o / a.D, d = o - f * a.D;
        const int i0v = a.i0[f];
        const double *wq = a.w + 4 * (size_t)f;
        const int r0 = i0v * a.D + d;
        if (OUT_F64 || d < a.nroot) {
            double c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double *e = a.Et64 + (r0 + j * a.D);
                double acc = a.mean[r0 + j * a.D];
                for (int k = 0; k < a.L; k++) acc = fma(e[(size_t)k * a.R], mg_load_lat<LAT_F64>(a.lat, b * a.ld + k), acc);
                c[j] = acc;
            }
            double v = wq[0] * c[0];
            v = fma(wq[1], c[1], v);
            v = fma(wq[2], c[2], v);
            v = fma(wq[3], c[3], v);
            if (OUT_F64) ((double *)a.out)[idx] = v;
            else ((float *)a.out)[idx] = (float)v;
        } else {
            float c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float *e = a.Et32 + (r0 + j * a.D);
                float acc = (float)a.mean[r0 + j * a.D];
                for (int k = 0; k < a.L; k++) acc = fmaf(e[(size_t)k * a.R], (float)mg_load_lat<LAT_F64>(a.lat, b * a.ld + k), acc);
                c[j] = acc;
            }
            float v = (float)wq[0] * c[0];
            v = fmaf((float)wq[1], c[1], v);
            v = fmaf((float)wq[2], c[2], v);
            v = fmaf((float)wq[3], c[3], v);
            ((float *)a.out)[idx] = v;
        }
    }
}

// Spline evaluation from explicit float64 coefficient arrays (n, NB, D) -> (n, T, D):
// MotionSpline.get_motion_vector / evaluate (reference motion_spline.py:71-92).
// sp = w0*c0, then fma in j order (splev.f sums j ascending).
__global__ __launch_bounds__(256) void mg_spline_eval_kernel(const double *coeffs, const int32_t *i0, const double *w,
                                                             double *out, int64_t n, int32_t T, int32_t D, int32_t R) {
    const int64_t TD = (int64_t)T * D, total = n * TD;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = idx / TD;
        const int o = (int)(idx - s * TD);
        const int f = o / D, d = o - f * D;
        const double *c = coeffs + s * R + (size_t)i0[f] * D + d;
        const double *wq = w + 4 * (size_t)f;
        double v = wq[0] * c[0];
        v = fma(wq[1], c[D], v);
        v = fma(wq[2], c[2 * D], v);
        v = fma(wq[3], c[3 * D], v);
        out[idx] = v;
    }
}

// -----------------------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------------------
struct mg_launch_events { hipEvent_t start = nullptr, stop = nullptr; };   // both NULL: an ordinary launch
template <int KK, bool LAT_F64, bool FUSE>
static int mg_launch_ws_inst(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a,
                             int buf_bytes, int lds, int grid, bool cs, const mg_launch_events &ev) {
    // hipExtLaunchKernelGGL with NULL events is hipLaunchKernelGGL; with events the dispatch records its own begin and end
    if (cs)
        hipExtLaunchKernelGGL((mg_frames_cs_kernel<KK, LAT_F64, FUSE>), dim3(grid), dim3(MG_CS_BLOCK), lds, p->ctx->stream, ev.start, ev.stop, 0,
                              (const float *)p->d_Epack, (const float *)p->d_mean32, (const double *)p->d_Erpack, (const double *)p->d_meanroot, lat,
                              (const int32_t *)g->d_i0, (const float4 *)g->d_w32, (const double *)g->d_wtap, (const mg_chunk *)g->d_chunks, out,
                              (const double *)p->d_gPpack, (const double *)p->d_gmPpad, (const double *)p->d_gconst, logp, a, (int)p->K,
                              (int)((p->L + 15) / 16), buf_bytes);
    else
        hipExtLaunchKernelGGL((mg_frames_ws_kernel<KK, LAT_F64, FUSE>), dim3(grid), dim3(MG_WS_BLOCK), lds, p->ctx->stream, ev.start, ev.stop, 0,
                              (const float *)p->d_Epack, (const float *)p->d_mean32, (const double *)p->d_Erpack, (const double *)p->d_meanroot, lat,
                              (const int32_t *)g->d_i0, (const float4 *)g->d_w32, (const double *)g->d_wtap, (const mg_chunk *)g->d_chunks, out,
                              (const double *)p->d_gPpack, (const double *)p->d_gmPpad, (const double *)p->d_gconst, logp, a, (int)p->K,
                              (int)((p->L + 15) / 16), buf_bytes);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

template <int KK>
static int mg_launch_ws_kk(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a,
                           bool lat_f64, int buf_bytes, int lds, int grid, bool cs, const mg_launch_events &ev) {
    if (logp) {
        // fused instances exist for <= 40 components: beyond that the mixture fragments no longer fit the
        // register budget next to the sweep (mg_frames_can_fuse_gmm refuses, so this is never reached)
        if constexpr (KK <= MG_FUSE_MAX_KK)
            return lat_f64 ? mg_launch_ws_inst<KK, true, true>(p, g, lat, out, logp, a, buf_bytes, lds, grid, cs, ev)
                           : mg_launch_ws_inst<KK, false, true>(p, g, lat, out, logp, a, buf_bytes, lds, grid, cs, ev);
        mg_set_error("mg_step_frames_and_logp: no fused kernel for %d components", p->L);
        return MG_ERR_UNSUPPORTED;
    }
    return lat_f64 ? mg_launch_ws_inst<KK, true, false>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, cs, ev)
                   : mg_launch_ws_inst<KK, false, false>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, cs, ev);
}

template <int KK>
static int mg_set_attr_kk() {
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if constexpr (KK <= MG_FUSE_MAX_KK) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    return MG_OK;
}

int mg_setup_kernel_attributes(mg_context *) {
    int rc;
#ifndef MG_ONLY_KK10
    if ((rc = mg_set_attr_kk<2>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<4>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<6>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<8>()) != MG_OK) return rc;
#endif
    if ((rc = mg_set_attr_kk<10>()) != MG_OK) return rc;
#ifndef MG_ONLY_KK10
    if ((rc = mg_set_attr_kk<12>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<14>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<16>()) != MG_OK) return rc;
#endif
    return MG_OK;
}

// LDS of the fused mixture scoring: two term buffers and two exp buffers of [K][16] float64
static int mg_fused_gmm_lds(const mg_primitive *p) { return 4 * p->K * 16 * 8; }

bool mg_frames_can_fuse_gmm(const mg_primitive *p, const mg_time_grid *g, int64_t B) {
    const int64_t n_tiles = (B + MG_NCAND - 1) / MG_NCAND;
    const int64_t grid = std::min<int64_t>(n_tiles * g->n_chunks, std::max(1, p->ctx->n_cu - p->ctx->reserved_cus));
    // fused: the mixture spans exactly the spatial latents
    return g->mfma_ok && p->d_gPpack != nullptr && p->K <= 16 && p->KK <= MG_FUSE_MAX_KK && p->Lg == p->L &&
           g->lds_bytes + mg_fused_gmm_lds(p) <= 160 * 1024 &&
           (n_tiles + grid - 1) / grid <= 4;   // the fused scoring handles at most four 16-candidate tiles per workgroup
}

// Which of the two LDS-staged kernels a launch over B candidates uses (1 = tile-major, 2 = chunk-stationary; -1 = the
// chunk-stationary one was asked for by option and does not cover the shape): the chunk-stationary one once every workgroup
// gets at least two units out of the one-time load of its chunk's eigenvector window ('walk', same box and buffer, us per
// step, tile-major / chunk-stationary: B = 1024 18.9 / 18.9, 2048 26.6 / 25.7, 4096 45.4 / 43.8, 6144 65.0 / 61.8,
// 8192 83.3 / 81.0 before the start-up work and 87.4 / 81-83 after it), the tile-major one for smaller batches.
int mg_frames_kernel_choice(const mg_primitive *p, const mg_time_grid *g, int64_t B, bool fused) {
    const int64_t n_tiles = (B + MG_NCAND - 1) / MG_NCAND;
    const int64_t units = n_tiles * g->n_chunks;
    const int64_t grid0 = std::min<int64_t>(units, std::max(1, p->ctx->n_cu - p->ctx->reserved_cus));
    const int64_t grid_cs = grid0 / g->n_chunks * g->n_chunks;   // whole workgroups per chunk
    const int want = p->ctx->opt[MG_OPT_FRAMES_KERNEL];
    bool cs = g->cs_ok && grid_cs >= g->n_chunks && grid_cs <= 4096 &&
              (!fused || (g->cs_lds_bytes + mg_fused_gmm_lds(p) <= 160 * 1024 && (n_tiles + grid_cs - 1) / grid_cs <= 4));
    if (want == 2 && !cs) return -1;
    if (want == 1) cs = false;
    else if (want == 0) cs = cs && units >= 2 * grid0;
    return cs ? 2 : 1;
}
int mg_frames_grid(const mg_primitive *p, const mg_time_grid *g, int64_t B, int which) {
    const int64_t units = (B + MG_NCAND - 1) / MG_NCAND * g->n_chunks;
    const int64_t grid0 = std::min<int64_t>(units, std::max(1, p->ctx->n_cu - p->ctx->reserved_cus));
    return (int)(which == 2 ? grid0 / g->n_chunks * g->n_chunks : grid0);
}

int mg_frames_lds_bytes(const mg_primitive *p, const mg_time_grid *g, int which, bool fused) {
    return (which == 2 ? g->cs_lds_bytes : g->lds_bytes) + (fused ? mg_fused_gmm_lds(p) : 0);
}

int mg_launch_frames_mfma(mg_primitive *p, const mg_time_grid *g, const void *lat, int ldt, int64_t B, int64_t ld, float *out, float *logp,
                          int prof_slot, int prof_slot2) {
    mg_frames_args a;
    a.B = B; a.ld = ld; a.T = g->T; a.D = p->D; a.Dp = p->Dp; a.cshift = p->cshift; a.L = p->L; a.nroot = p->nroot;
    a.n_chunks = g->n_chunks; a.stride = g->stride; a.max_wi = g->max_wi; a.max_nt = g->max_nt; a.nbuf = g->nbuf;
    a.debug = 0;
#ifdef MG_DEBUG_BUILD
    if (const char *dbg_env = getenv("MG_DEBUG_FLAGS")) a.debug = atoi(dbg_env);   // read per launch: A/B tools switch it inside one process
#endif
    const int64_t n_tiles = (B + MG_NCAND - 1) / MG_NCAND;
    const int64_t units = n_tiles * g->n_chunks;
    if (n_tiles >= ((int64_t)1 << 27) || units >= ((int64_t)1 << 31)) {
        mg_set_error("mg_back_project_frames: batch too large for one launch");
        return MG_ERR_UNSUPPORTED;
    }
    a.n_tiles = (int32_t)n_tiles;
    a.max_tiles = g->max_tiles;
    for (int i = 0; i < MG_ARG_CHUNKS; i++) a.ck[i] = i < g->n_chunks ? g->chunks[i] : mg_chunk{};
    a.cs_magic = a.cs_per = a.cs_rem = 0;
    const bool lf = (ldt == MG_F64);
    const int which = mg_frames_kernel_choice(p, g, B, logp != nullptr);
    if (which < 0) {
        mg_set_error("mg_back_project_frames: the chunk-stationary kernel does not cover this shape (window of %d row tiles, %d bytes of LDS)",
                     g->max_tiles, g->cs_lds_bytes);
        return MG_ERR_UNSUPPORTED;
    }
    const bool cs = which == 2;
    // nbuf ring slots (image + root outputs + tables), the float64 root image, the progress counters
    const int buf_bytes = (MG_NCAND * g->stride * 4 + 255) / 256 * 256;
    int lds = g->nbuf * (buf_bytes + MG_RO_BYTES_N(g->max_nt) + MG_TB_BYTES_N(g->max_nt)) + MG_NCAND * (g->max_wi * p->nroot + 1) * 8 + 128;
    if (lds != g->lds_bytes || lds > 160 * 1024 || (logp && !mg_frames_can_fuse_gmm(p, g, B))) {
        mg_set_error("mg_back_project_frames: internal LDS sizing mismatch (%d vs %d)", lds, g->lds_bytes);
        return MG_ERR_UNSUPPORTED;
    }
    if (cs) lds = g->cs_lds_bytes;
    if (logp) lds += mg_fused_gmm_lds(p);
    const int grid = mg_frames_grid(p, g, B, which);
    if (cs) {
        const int Q = grid / g->n_chunks;
        a.cs_magic = (1 << 20) / g->n_chunks + 1;
        a.cs_per = (int32_t)(n_tiles / Q);
        a.cs_rem = (int32_t)(n_tiles % Q);
    }
    mg_launch_events ev;
    if (prof_slot >= 0) (void)mg_prof_kernel(p->ctx, prof_slot, prof_slot2, &ev.start, &ev.stop);
    switch (p->KK) {
#ifndef MG_ONLY_KK10
        case 2: return mg_launch_ws_kk<2>(p, g, lat, out, logp, a, lf, buf_bytes, lds, grid, cs, ev);
        case 4: return mg_launch_ws_kk<4>(p, g, lat, out, logp, a, lf, buf_bytes, lds, grid, cs, ev);
        case 6: return mg_launch_ws_kk<6>(p, g, lat, out, logp, a, lf, buf_bytes, lds, grid, cs, ev);
        case 8: return mg_launch_ws_kk<8>(p, g, lat, out, logp, a, lf, buf_bytes, lds, grid, cs, ev);
#endif
        case 10: return mg_launch_ws_kk<10>(p, g, lat, out, logp, a, lf, buf_bytes, lds, grid, cs, ev);
#ifndef MG_ONLY_KK10
        case 12: return mg_launch_ws_kk<12>(p, g, lat, out, logp, a, lf, buf_bytes, lds, grid, cs, ev);
        case 14: return mg_launch_ws_kk<14>(p, g, lat, out, logp, a, lf, buf_bytes, lds, grid, cs, ev);
        case 16: return mg_launch_ws_kk<16>(p, g, lat, out, logp, a, lf, buf_bytes, lds, grid, cs, ev);
#endif
        default: mg_set_error("mg_back_project_frames: MFMA path needs n_components <= 64"); return MG_ERR_UNSUPPORTED;
    }
}

int mg_launch_frames_direct(mg_primitive *p, const mg_time_grid *g, const void *lat, int ldt, int64_t B, int64_t ld, void *out, bool out_f64) {
    mg_direct_args a;
    a.Et32 = p->d_Et32; a.Et64 = p->d_Et64; a.mean = p->d_mean; a.lat = lat; a.i0 = g->d_i0; a.w = g->d_w; a.out = out;
    a.B = B; a.ld = ld; a.T = g->T; a.D = p->D; a.L = p->L; a.R = p->R; a.nroot = p->nroot;
    int64_t total = B * (int64_t)g->T * p->D;
    int64_t blocks = (total + 255) / 256;
    int grid = (int)std::min<int64_t>(blocks, (int64_t)p->ctx->n_cu * 32);
    hipStream_t st = p->ctx->stream;
    const bool lf = (ldt == MG_F64);
    if (lf && out_f64) hipLaunchKernelGGL((mg_frames_direct_kernel<true, true>), dim3(grid), dim3(256), 0, st, a);
    else if (lf) hipLaunchKernelGGL((mg_frames_direct_kernel<true, false>), dim3(grid), dim3(256), 0, st, a);
    else if (out_f64) hipLaunchKernelGGL((mg_frames_direct_kernel<false, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((mg_frames_direct_kernel<false, false>), dim3(grid), dim3(256), 0, st, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

int mg_launch_spline_eval(mg_primitive *p, const mg_time_grid *g, const double *coeffs, int64_t n, double *out) {
    int64_t total = n * (int64_t)g->T * p->D;
    int grid = (int)std::min<int64_t>((total + 255) / 256, (int64_t)p->ctx->n_cu * 32);
    hipLaunchKernelGGL(mg_spline_eval_kernel, dim3(grid), dim3(256), 0, p->ctx->stream, coeffs, g->d_i0, g->d_w, out, n, g->T, p->D, p->R);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

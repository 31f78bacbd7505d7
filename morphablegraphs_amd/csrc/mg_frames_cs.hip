// The chunk-stationary LDS-staged frames kernel (gfx950): the hot path of the bench's step.
#include "mg_frames_common.h"

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
#ifndef MG_CS_GMM_EARLY
#define MG_CS_GMM_EARLY 0   // 1: the fused mixture's first group of tiles is scored while the pipeline fills (below), not in the tail -- A/B only: SLOWER
#endif
#ifndef MG_CS_STORE_THROTTLE
#define MG_CS_STORE_THROTTLE -1   // >= 0 (A/B only): the sweep waits after each trip of its lean loop until at most so many of the wave's stores are in flight
#endif
#ifndef MG_CS_THROTTLE_LAST_ONLY
#define MG_CS_THROTTLE_LAST_ONLY 1   // ... in the workgroup's last unit only (while the mixture's loads share the CU's memory pipe with it)
#endif
#ifndef MG_CS_QUAD_FRAGMENTS
#define MG_CS_QUAD_FRAGMENTS 1   // the one-time eigenvector fragment loads from the quad copy of the image: 16-byte loads
#endif
#ifndef MG_CS_GMM_EARLY_HALF
#define MG_CS_GMM_EARLY_HALF 1   // with the staged tail: components 4 .. 7 are scored at start-up by the four sweep waves that produce nothing
#endif
#ifndef MG_CS_GMM_LDSX
#define MG_CS_GMM_LDSX 1   // float32 latents: the mixture's two latent tiles staged in LDS at start-up, both components of a producer wave requested at once
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
// the staged latent tiles of the fused mixture: behind its term and exp buffers ([4][K*16] float64 after the 64 counters)
__device__ __forceinline__ mg_lds_f32 *mg_cs_gmm_x(mg_lds_int *prog, int gK) { return (mg_lds_f32 *)((mg_lds_f64 *)(prog + MG_CS_PROG_INTS) + 4 * gK * 16); }
__device__ __forceinline__ mg_lds_f64 *mg_cs_gmm_mp(mg_lds_int *prog, int gK, int KK) { return (mg_lds_f64 *)(mg_cs_gmm_x(prog, gK) + 2 * KK * 64); }   // [K][JT*16], then [K]
// MG_CS_GMM_EARLY_HALF: are components 4 .. 7 scored at start-up?  Where the workgroup has BOTH tiles of a mixture group (uniform over the workgroup; the
// start-up waves and the tail ask the same question).  Measured, one process and one buffer each: B = 8192 -0.3 us (four boxes: +0.3, -0.7, -0.5, -0.3),
// 12 000 -0.9, 16 384 -1.0 -- and with ONE tile per workgroup +0.6 (B = 2048) / +0.9 us (4096): the start-up pays the same and the tail has half to gain.
__device__ __forceinline__ bool mg_cs_gmm_early(const mg_frames_args &a) {
    const int64_t gt0 = (int64_t)blockIdx.x * a.n_tiles / gridDim.x, gt1 = ((int64_t)blockIdx.x + 1) * a.n_tiles / gridDim.x;
    return a.gmm_staged && gt0 + 1 < gt1;
}
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
__device__ __forceinline__ void mg_cs_load_fragments(float (&ef)[NT][KK], const float2 *ep, const mg_chunk &ck, int first, int step, int lane, [[maybe_unused]] int rt_total) {
#if MG_CS_QUAD_FRAGMENTS
    // from the quad copy behind the pair image (mg_host.hip): per tile [KK / 4][64][4] floats, then [64][2] for the last pair: 16-byte loads
    const float *eq = (const float *)ep + (size_t)rt_total * KK * 64;
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int t = first + step * i;
        const int tc = t < ck.ntiles ? t : ck.ntiles - 1;   // clamp: redundant but in bounds
        const float *p = eq + (size_t)(ck.rt0 + tc) * KK * 64;
#pragma unroll
        for (int q = 0; q < KK / 4; q++) {
            const f32x4 v = *(const f32x4 *)(p + ((size_t)q * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; e++) ef[i][4 * q + e] = v[e];
        }
        if constexpr (KK % 4 == 2) {
            const float2 v = *(const float2 *)(p + (size_t)(KK / 4) * 256 + (size_t)lane * 2);
            ef[i][KK - 2] = v.x;
            ef[i][KK - 1] = v.y;
        }
    }
#else
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
#endif
}

template <int KK, bool LAT_F64, bool FUSE_GMM, bool SPLIT>
__global__ __launch_bounds__(MG_CS_BLOCK) void mg_frames_cs_kernel(
    const float *__restrict__ Epack,      // [RT][KK/2][64][2]
    const float *__restrict__ mean32,     // [RT*16]
    const double *__restrict__ Erpack,    // [RRT][KK][64]
    const double *__restrict__ meanroot,  // [RRT*16]
    const void *__restrict__ lat,         // (B, ld) f32 or f64
    const int32_t *__restrict__ i0tab,    // (T)
    const float4 *__restrict__ w32,       // (T)
    const double *__restrict__ wtap,      // [n_chunks][FT][KS][64]
    const float4 *__restrict__ rootm,     // (T, 2): SPLIT: the root channels' mean part {Mhi, Mlo}
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
    MG_LITE_DECL
    MG_LITE(0);
    constexpr bool GMM_LDSX = FUSE_GMM && !LAT_F64 && MG_CS_GMM_LDSX && !MG_CS_GMM_EARLY && MG_CS_NCW - MG_CS_NSP == 4;   // (four staging waves)
    constexpr bool GMM_HALF = GMM_LDSX && MG_CS_GMM_EARLY_HALF && MG_WS_NPW == 4;
    if (wave == 0) MG_SUB_STAMP(14, 0, 0);
    // Kernel arguments: left alone, the compiler fetches each where a role first uses it -- a dependent trip to memory per
    // 64-byte line of the argument block (wave 0 alone made eight in a row before it had issued its loads, 4.8 us after entry).
    // Asking for all of them here makes that one trip; the later uses hit the scalar cache.
    asm volatile("" ::"s"(Epack), "s"(mean32), "s"(Erpack), "s"(meanroot), "s"(lat), "s"(i0tab), "s"(w32), "s"(wtap), "s"(rootm), "s"(out), "s"(gPpack),
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
        // ... and two of them stage the latent tiles of the workgroup's mixture scoring (MG_CS_GMM_LDSX, float32 latents): requested with mean'
        [[maybe_unused]] float gxv[KK];
        [[maybe_unused]] const int gx_tile = cj - MG_CS_NSP;   // 0, 1: the group's first / second tile
        [[maybe_unused]] bool gx_on = false;
        [[maybe_unused]] double gmv[6];
        [[maybe_unused]] mg_gmm_frag<KK> epf;
        [[maybe_unused]] bool early_on = false;
        if (!producing) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int e = e0 + i * MEAN_NTH;
                mv[i] = mean32[(size_t)ck.rt0 * 16 + (e < n_mean ? e : n_mean - 1)];
            }
            if constexpr (GMM_LDSX) {
                const int64_t gt0 = (int64_t)blockIdx.x * a.n_tiles / gridDim.x, gt1 = ((int64_t)blockIdx.x + 1) * a.n_tiles / gridDim.x;
                gx_on = a.gmm_staged && gx_tile < 2 && gt0 + gx_tile < gt1;   // (wave-uniform)
                if (gx_on) {
                    const int64_t b0 = (gt0 + gx_tile) * MG_NCAND;
                    mg_gmm_load_x<KK, false>(gxv, lat, b0, (int)((a.B - b0) < MG_NCAND ? (a.B - b0) : MG_NCAND), a.ld, L, cl, g);
                }
                const int n_mp = gK * gJT * 16;              // the third such wave: the components' C-in rows; the fourth: their constants
                if (!a.gmm_staged) {
                } else if (gx_tile == 2) {
#pragma unroll
                    for (int i = 0; i < 6; i++) gmv[i] = gmP[min(lane + 64 * i, n_mp - 1)];
                } else if (gx_tile == 3) {
                    gmv[0] = gcst[min(lane, gK - 1)];
                }
                if constexpr (GMM_HALF) {
                    early_on = mg_cs_gmm_early(a) && 4 + gx_tile < gK;
                    if (early_on) mg_gmm_load_pf2<KK>(epf, gPpack, gK, 4 + gx_tile, gJT, lane);
                }
            }
        }
        mg_lds_barrier();
        if (producing) {
            mg_cs_load_fragments<KK, TPWS>(ef, ep, ck, nt_p + cj, MG_CS_NSP, lane, a.rt_total);   // in flight across barrier B
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (e0 + i * MEAN_NTH < n_mean) lds_mean[e0 + i * MEAN_NTH] = mv[i];
            for (int e = e0 + 6 * MEAN_NTH; e < n_mean; e += MEAN_NTH) lds_mean[e] = mean32[(size_t)ck.rt0 * 16 + e];   // (windows beyond 6 * MEAN_NTH rows)
            mg_publish(prog + MG_CS_PROG_MEAN, cj - MG_CS_NSP, lane, 1);   // (the first unit waits for this: nothing of the mixture's in front of it)
            MG_LITE(5);
            if constexpr (GMM_LDSX) {   // the mixture's tables, with a flag of their own (prog + MG_CS_PROG_GMM + 20 + i) that the tail waits for
                if (gx_on) {
                    mg_lds_f32 *gx = mg_cs_gmm_x(prog, gK) + gx_tile * KK * 64;
#pragma unroll
                    for (int kk = 0; kk < KK; kk++) gx[kk * 64 + lane] = gxv[kk];
                }
                const int n_mp = gK * gJT * 16;
                mg_lds_f64 *mpl = mg_cs_gmm_mp(prog, gK, KK);
                if (!a.gmm_staged) {
                } else if (gx_tile == 2) {
#pragma unroll
                    for (int i = 0; i < 6; i++)
                        if (lane + 64 * i < n_mp) mpl[lane + 64 * i] = gmv[i];
                    for (int e = lane + 64 * 6; e < n_mp; e += 64) mpl[e] = gmP[e];   // (more than 384 entries: K * JT > 24)
                } else if (gx_tile == 3) {
                    if (lane < gK) mpl[n_mp + lane] = gmv[0];
                }
                if (a.gmm_staged) mg_publish(prog + MG_CS_PROG_GMM + 20, gx_tile, lane, 1);
                if constexpr (GMM_HALF) {
                    if (early_on) {
                        __builtin_amdgcn_s_setprio(0);   // (below wave 0's root chains and everything else of the first unit)
                        mg_wait_producers(prog + MG_CS_PROG_GMM + 20, 1);   // all four waves' tables are in LDS
                        mg_fused_gmm_early_component<KK>(prog + MG_CS_PROG_GMM, epf, 4 + gx_tile, mg_cs_gmm_x(prog, gK), mg_cs_gmm_mp(prog, gK, KK),
                                                         mg_cs_gmm_mp(prog, gK, KK) + gK * gJT * 16, a.n_tiles, gK, gJT, lane);
                        __builtin_amdgcn_s_setprio(3);
                    }
                    if (a.gmm_staged) mg_publish(prog + MG_CS_PROG_GMM + 28, gx_tile, lane, 1);   // this wave's early terms (if any) are written
                }
            }
        }
        if constexpr (FUSE_GMM && MG_CS_GMM_EARLY) {
            // Round 5's structural attempt, kept for A/B (tools/build_variant.sh x -DMG_CS_GMM_EARLY=1): the mixture while the pipeline
            // fills.  In the tail (the default) the mixture costs what a launch of its own costs -- the frames alone 69.3 us, the mixture
            // kernel alone 11.2 us, fused 78.5-82 us: four waves' chains of float64 MFMAs behind 90 KB of fragment loads that wait in
            // the CU's memory queue behind the last unit's stores.  The four sweep waves that produce nothing are idle until the first
            // unit is in LDS (~8 us after entry, the HBM idle): here they score the workgroup's first two tiles then, at the lowest
            // priority.  MEASURED SLOWER: 83.9 against 78.5 us in one process on one buffer (tools/ab.py) -- the fragment loads delay the
            // row fragments, the float64 MFMAs the first unit, and every microsecond of the first unit is a microsecond of the kernel.
            // (Two more attempts at the tail's round trips: both components of a wave requested at once, MG_GMM_PAIR_LOADS: 73 spilled
            // registers, 89.8 us; a component held in the idle sweep waves' registers from the start: 38 spilled, 87.4 us.)
            if (!producing) {
                __builtin_amdgcn_s_setprio(0);
                mg_lds_int *gprog = prog + MG_CS_PROG_GMM;
                const int gw = cj - MG_CS_NSP;
                mg_fused_gmm_terms<KK, LAT_F64>(gprog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, gw, lane, 0);
                if (gw < 2) mg_fused_gmm_finish(gprog, logp, a.B, a.n_tiles, gK, gw, lane, 0);
                __builtin_amdgcn_s_setprio(3);
            }
        }
        const int nql = (D - nroot + 3) >> 2;             // quad lanes per sample
        const int gl = nql + 1;                           // + the root lane
        const int rpi = 64 / gl;                          // samples per wave instruction
        // Three samples of 20 lanes: the third one sits in lanes 44 .. 63, not 40 .. 59.  ds_read_b128 serves the lanes in four
        // groups of sixteen ({36-43, 48-51, 60-63} is one), and a row of 80 floats wraps around the 64 banks: the second sample's
        // last quads (columns 68 .. 79, lanes 36-38) and the third sample's first ones (columns 4 .. 15, lanes 40-42) are different
        // addresses in the same banks whenever the two samples share their tap rows -- one extra LDS cycle on most tap reads.
        // Shifted, lanes 60-62 read the addresses lanes 36-38 read (a broadcast).
        const bool shift3 = MG_SWEEP_LANEMAP && gl == 20;
        const int lane_s = (shift3 && lane >= 40) ? lane - 4 : lane;
        const int fsub = lane_s / gl, ql = lane_s - fsub * gl;
        const bool lane_on = shift3 ? (lane < 40 || lane >= 44) : lane < rpi * gl;
        const bool root_lane = ql == nql;
        const int d0 = root_lane ? 0 : nroot + 4 * ql;    // first channel of this lane
        const int nst = root_lane ? nroot : (D - d0 < 4 ? D - d0 : 4);
        const int64_t TD = (int64_t)a.T * D;
        const int dp4 = Dp * 4;
        // byte offset of the lane's quad inside a basis row; SPLIT: the root lane takes the row's first quad, {padding, root channels}
        // (float64 pipeline: the root lane reads what its row's first quad lane reads -- a broadcast -- and so holds channel 3, the
        // fourth float of its store, without asking another lane for it)
        const int lane_img = (SPLIT && root_lane) ? 0 : ((root_lane ? nroot : d0) + a.cshift) * 4;
        const float *lds_m = (const float *)ro_base;    // SPLIT: {Mhi, Mlo} per sample of the chunk (where the root outputs would be)
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
            if (u == 0) MG_LITE(1);
            if (u == n_units - 1) MG_LITE(2);
            const int slot = u & 1;
            [[maybe_unused]] const bool throttle_now = FUSE_GMM && u == n_units - 1;
            const unsigned char *img = smem + (size_t)slot * buf_bytes;
            const float *lds_ro = (const float *)(ro_base + (size_t)slot * RO_BYTES);
            const bool has1 = cj + 8 < ncand;
            const int c1 = has1 ? cj + 8 : cj;
            const unsigned char *img0 = img + (size_t)(cj * stride + col0) * 4 + lane_img;
            const unsigned char *img1 = img + (size_t)(c1 * stride + col0) * 4 + lane_img;
            const float *ro0 = lds_ro + cj * MG_RO_CS(max_nt), *ro1 = lds_ro + c1 * MG_RO_CS(max_nt);
            float *or0 = out + (size_t)(b0 + cj) * TD + (size_t)ck.t0 * D;   // wave-uniform row bases
            float *or1 = out + (size_t)(b0 + c1) * TD + (size_t)ck.t0 * D;
            auto sweep_rows = [&](auto pitch_tag, auto all4_tag, int f_first, int f_last) {
                constexpr int DP4 = decltype(pitch_tag)::value;
                constexpr bool ALL4 = decltype(all4_tag)::value;   // the usual shape as a constant: every lane stores four floats
                const bool all4l = ALL4 ? true : all4;
                if constexpr (ALL4 && !SPLIT && MG_SWEEP_FAST) {
                    // The trips whose six samples all lie inside the chunk, lean: the lanes that hold no sample sit the whole loop
                    // out (one exec mask for the loop, none per store); every lane runs the taps -- the root lane on the first
                    // quad lane's columns, so that channel 3 is already in its first register: no cross-lane read -- and the root
                    // lanes then take the three root outputs from wave 0's table; the stores address memory as a scalar base +
                    // a 32-bit lane offset.  Same operations on the same values as the general loop below: the same bits.
                    if (MG_DBG(4 | 8192 | 16384)) {   // (the sweep's ablations live in the general loop)
                    } else if (lane_on) {
                        for (; f_first + 2 * rpi <= f_last; f_first += 2 * rpi) {
                            const int fa_ = f_first + fsub, fb_ = fa_ + rpi;
                            const float4 wa = lds_w[fa_], wb = lds_w[fb_];
                            const int moa = lds_mo[fa_], mob = lds_mo[fb_];
                            const mg_tap_rows r0a = mg_quad_load<DP4>(img0 + moa, dp4), r0b = mg_quad_load<DP4>(img0 + mob, dp4);
                            const mg_tap_rows r1a = mg_quad_load<DP4>(img1 + moa, dp4), r1b = mg_quad_load<DP4>(img1 + mob, dp4);
                            __builtin_amdgcn_sched_barrier(0);
                            f32x4 v0a = mg_quad_fma(r0a, wa), v0b = mg_quad_fma(r0b, wb);
                            f32x4 v1a = mg_quad_fma(r1a, wa), v1b = mg_quad_fma(r1b, wb);
                            if (root_lane) {
                                v0a = mg_root_merge(v0a, ro0 + fa_ * 4); v0b = mg_root_merge(v0b, ro0 + fb_ * 4);
                                v1a = mg_root_merge(v1a, ro1 + fa_ * 4); v1b = mg_root_merge(v1b, ro1 + fb_ * 4);
                            }
                            float *pa0 = or0 + (size_t)f_first * D, *pa1 = or1 + (size_t)f_first * D;          // uniform
                            mg_store4_s(pa0, lane_out_b, v0a);
                            mg_store4_s(pa0 + (size_t)rpi * D, lane_out_b, v0b);
                            if (has1) {
                                mg_store4_s(pa1, lane_out_b, v1a);
                                mg_store4_s(pa1 + (size_t)rpi * D, lane_out_b, v1b);
                            }
                            if constexpr (MG_CS_STORE_THROTTLE >= 0) {   // A/B only: at most so many stores of this wave in flight (see the macro)
                                if (!MG_CS_THROTTLE_LAST_ONLY || throttle_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MG_CS_STORE_THROTTLE) : "memory");
                            }
                        }
                    } else {
                        f_first += (f_last - f_first) / (2 * rpi) * (2 * rpi);
                    }
                }
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
                        } else if (SPLIT || !root_lane) {   // all 16 tap rows are requested before the first FMA
                            const float4 wa = lds_w[fa_], wb = lds_w[fb_];
                            const int moa = lds_mo[fa_], mob = lds_mo[fb_];
                            mg_rootm ma, mb;   // SPLIT: every lane asks (a broadcast read), the root lanes use them
                            if constexpr (SPLIT) { ma = mg_rootm_load(lds_m, fa_); mb = mg_rootm_load(lds_m, fb_); }
                            mg_tap_rows r0a, r0b, r1a, r1b;
                            if (MG_DBG(16384)) {   // ablation: no tap reads (the FMAs run on what is in registers anyway)
                                const f32x4 k = {wa.x, wa.y, wb.z, wb.w};
                                r0a = r0b = r1a = r1b = mg_tap_rows{k, k, k, k};
                            } else {
                                r0a = mg_quad_load<DP4>(img0 + moa, dp4); r0b = mg_quad_load<DP4>(img0 + mob, dp4);
                                r1a = mg_quad_load<DP4>(img1 + moa, dp4); r1b = mg_quad_load<DP4>(img1 + mob, dp4);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (MG_DBG(8192)) {   // ablation: the tap reads, no FMAs
                                v0a = r0a.t0 + r0a.t1 * 0.f; v0b = r0b.t2; v1a = r1a.t3; v1b = r1b.t1;
                                asm volatile("" ::"v"(r0a.t2), "v"(r0a.t3), "v"(r0b.t0), "v"(r0b.t1), "v"(r0b.t3), "v"(r1a.t0), "v"(r1a.t1), "v"(r1a.t2), "v"(r1b.t0), "v"(r1b.t2), "v"(r1b.t3));
                            } else {
                                v0a = mg_quad_fma(r0a, wa);
                                v0b = mg_quad_fma(r0b, wb);
                                v1a = mg_quad_fma(r1a, wa);
                                v1b = mg_quad_fma(r1b, wb);
                            }
                            if constexpr (SPLIT) {
                                if (root_lane) {
                                    if constexpr (ALL4) {   // (all4: three root channels in columns 1 .. 3)
                                        v0a = mg_root_finish<1>(v0a, ma); v0b = mg_root_finish<1>(v0b, mb);
                                        v1a = mg_root_finish<1>(v1a, ma); v1b = mg_root_finish<1>(v1b, mb);
                                    } else {
                                        v0a = mg_root_finish_rt(v0a, ma, a.cshift); v0b = mg_root_finish_rt(v0b, mb, a.cshift);
                                        v1a = mg_root_finish_rt(v1a, ma, a.cshift); v1b = mg_root_finish_rt(v1b, mb, a.cshift);
                                    }
                                }
                            }
                        } else {
                            v0a = *(const f32x4 *)&ro0[fa_ * 4];
                            v0b = *(const f32x4 *)&ro0[fb_ * 4];
                            v1a = *(const f32x4 *)&ro1[fa_ * 4];
                            v1b = *(const f32x4 *)&ro1[fb_ * 4];
                        }
                        if (all4l && MG_DBG(4)) {
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (ob) mg_store4_at(pb0, lane_out_b, v0b);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                            if (ob && has1) mg_store4_at(pb1, lane_out_b, v1b);
                        } else if (all4l) {
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
                        if (SPLIT || !root_lane) {
                            const float4 wa = lds_w[fa_];
                            const int moa = lds_mo[fa_];
                            mg_rootm ma;
                            if constexpr (SPLIT) ma = mg_rootm_load(lds_m, fa_);
                            v0a = mg_quad_taps_t<DP4>(img0 + moa, wa, dp4);
                            v1a = mg_quad_taps_t<DP4>(img1 + moa, wa, dp4);
                            if constexpr (SPLIT) {
                                if (root_lane) {
                                    if constexpr (ALL4) { v0a = mg_root_finish<1>(v0a, ma); v1a = mg_root_finish<1>(v1a, ma); }
                                    else { v0a = mg_root_finish_rt(v0a, ma, a.cshift); v1a = mg_root_finish_rt(v1a, ma, a.cshift); }
                                }
                            }
                        } else {
                            v0a = *(const f32x4 *)&ro0[fa_ * 4];
                            v1a = *(const f32x4 *)&ro1[fa_ * 4];
                        }
                        if (all4l) {
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
                if (dp4 == 320 && all4) sweep_rows(std::integral_constant<int, 320>{}, std::true_type{}, 0, ck.nT);
                else sweep_rows(std::integral_constant<int, 0>{}, std::false_type{}, 0, ck.nT);
            }
            MG_STAMP(4);
            MG_UNIT_STAMP(u, 1);
            if (u == n_units - 1) MG_LITE(4);
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
        MG_LITE_DUMP;
    } else if (wave != 0) {
        // ================= row producers (waves 1..3): TPWP row tiles each, the fragments in registers =================
        const int pw = wave - 1;
        float ef[TPWP][KK];
        mg_lds_barrier();
        MG_LITE(5);
        MG_SUB_STAMP(14 + (wave == 1 ? 1 : 0), wave == 1 ? 0 : 1, 0);
        mg_cs_load_fragments<KK, TPWP>(ef, ep, ck, pw, NRP, lane, a.rt_total);   // in tile order: the first unit's MFMAs start as its first fragments land
        if (MG_DBG(32)) {   // when do the fragments land?
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MG_SUB_STAMP(14 + (wave == 1 ? 1 : 0), wave == 1 ? 0 : 1, 1);
        }
        mg_cs_wait_mean(prog);
        MG_LITE(2);
        MG_STAMP_DECL
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            if (u >= 2) mg_cs_wait_swept(prog, u - 1);   // the slot's previous unit has been swept
            mg_cs_wait_latents(prog, u + 1);
            MG_STAMP(5);
            MG_UNIT_STAMP(u, 0);
            if (u == 0) MG_LITE(6);
            if (!MG_DBG(1))
                mg_cs_produce<KK, TPWP>(ef, lds_latb + (u & 1) * KK * 64, lds_mean, (float *)(smem + (size_t)(u & 1) * buf_bytes), stride, pw, NRP, nt_p,
                                        lane, cl, g);
            MG_STAMP(2);
            MG_UNIT_STAMP(u, 1);
            mg_publish(prog, wave, lane, u + 1);
            if (u == 0) MG_LITE(7);
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
        if constexpr (SPLIT) {
            // the mean/delta split: no root stage.  Wave 0 parks the chunk's per-sample tables (tap weights, tap row offsets, the
            // root channels' {Mhi, Mlo}) once and stages a latent tile per unit; the root channels' deltas are rows of the ordinary
            // float32 tiles and the sweep's root lanes finish them.
            const int tl = lane < ck.nT ? lane : ck.nT - 1;   // clamped, unconditional
            const float4 tw_v = w32[ck.t0 + tl];
            const int ti_v = i0tab[ck.t0 + tl];
            const float4 mh_v = rootm[2 * (ck.t0 + tl)], ml_v = rootm[2 * (ck.t0 + tl) + 1];
            mg_lds_barrier();
            {
                float4 *tw = (float4 *)tb_base;
                int *tmo = (int *)(tw + max_nt);
                float4 *tm = (float4 *)ro_base;
                if (lane < ck.nT) {
                    tw[lane] = tw_v;
                    tmo[lane] = (ti_v - ck.imin) * Dp * 4;   // byte offset of the first tap row in the f32 image
                    tm[2 * lane] = mh_v;
                    tm[2 * lane + 1] = ml_v;
                }
            }
            MG_STAMP_DECL
            for (int u = 0; u < n_units; u++) {
                MG_STAMP(0);
#pragma unroll
                for (int kk = 0; kk < KK; kk++) s64frag[kk] = s64next[kk];
                if (u >= 2) mg_cs_wait_swept(prog, u - 1);   // every wave is done with the slot's previous unit, its latent tile included
                MG_STAMP(5);
                MG_UNIT_STAMP(u, 0);
                float *lb = lds_latb + (u & 1) * KK * 64;
#pragma unroll
                for (int kk = 0; kk < KK; kk++) lb[kk * 64 + lane] = (float)s64frag[kk];
                mg_publish(prog, MG_CS_PROG_LAT, lane, u + 1);
                load_latents(s64next, u + 1);   // a unit ahead
                MG_STAMP(3);
                MG_UNIT_STAMP(u, 1);
                mg_publish(prog, wave, lane, u + 1);   // (the first one: the tables are in LDS)
                MG_STAMP(4);
            }
            MG_STAMP_DUMP;
        } else {
        const double *rpp[3];
        double rm_v[3][4];   // mean of the root rows: the C-in of the root chains, parked in LDS
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int tc = t < ck.nrt ? t : ck.nrt - 1;
            rpp[t] = Erpack + ((size_t)(ck.rrt0 + tc) * KK) * 64 + lane;
        }
        // the root rows' float64 fragments stay in registers as well where the budget allows (6 KK VGPRs)
        constexpr bool RR = KK <= (LAT_F64 ? 8 : 10);   // (float64 latents double wave 0's latent-tile registers: the root fragments stream then)
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
        MG_LITE(5);
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
            tap_o_off[ct] = colok ? cc * MG_RO_CS(max_nt) + cd : 3;
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
            if (u == 0) MG_LITE(6);
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
                // the root rows' fragments streamed from L2 in groups of CH k-steps (a half of them up to 12 k-steps; four at a time
                // beyond: 3 x KK / 2 float64 fragments beside wave 0's other k-step arrays spilled 14-29 registers inside the unit loop
                // for 53 .. 64 latents -- round 4's code-object metadata)
                constexpr int CH = KK <= 12 ? KK / 2 : 4;
#pragma unroll
                for (int h0 = 0; h0 < KK; h0 += CH) {
                    double rp[3][CH];
#pragma unroll
                    for (int t = 0; t < 3; t++)
#pragma unroll
                        for (int q2 = 0; q2 < CH; q2++) rp[t][q2] = rpp[t][(h0 + q2 < KK ? h0 + q2 : KK - 1) * 64];
#pragma unroll
                    for (int q2 = 0; q2 < CH; q2++)
#pragma unroll
                        for (int t = 0; t < 3; t++)
                            if (h0 + q2 < KK) racc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rp[t][q2], (double)s64frag[h0 + q2 < KK ? h0 + q2 : KK - 1], racc[t], 0, 0, 0);
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
            if (u == 0) MG_LITE(7);
            mg_publish(prog, wave, lane, u + 1);
            if (FIRST) load_latents(s64next, 1);
            MG_STAMP(4);
        };
        if (n_units > 0) root_unit(0, std::true_type{});
        for (int u = 1; u < n_units; u++) root_unit(u, std::false_type{});
        MG_STAMP_DUMP;
        }
    }
    if (FUSE_GMM && wave < MG_WS_NPW) {   // the mixture: the four producer waves, as in the tile-major kernel
        mg_lds_int *gprog = prog + MG_CS_PROG_GMM;
        if constexpr (!MG_CS_GMM_EARLY) {
            MG_UNIT_STAMP(29, 0);   // (diagnostic build: the tail's timeline in rows "unit" 29 .. 30 of the producer waves)
            MG_LITE(1);
            bool staged = false;
            if constexpr (GMM_LDSX) staged = a.gmm_staged != 0;   // (uniform: where the staged tables did not fit LDS the tail loads everything itself)
            if (staged) {
                if constexpr (GMM_LDSX) {
                    mg_wait_producers(gprog + 20, 1);   // the four staging waves' flags (long since set)
                    mg_fused_gmm_terms_ldsx<KK>(gprog, gPpack, mg_cs_gmm_x(prog, gK), mg_cs_gmm_mp(prog, gK, KK), mg_cs_gmm_mp(prog, gK, KK) + gK * gJT * 16,
                                                a.n_tiles, gK, gJT, wave, lane, GMM_HALF && mg_cs_gmm_early(a));
                    if constexpr (GMM_HALF) mg_wait_producers(gprog + 28, 1);   // the start-up's terms (set long ago)
                }
            } else
                mg_fused_gmm_terms<KK, LAT_F64>(gprog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, wave, lane, 0);
            MG_UNIT_STAMP(29, 1);
            MG_LITE(3);
            MG_UNIT_STAMP(30, 0);
            if (wave < 2) mg_fused_gmm_finish(gprog, logp, a.B, a.n_tiles, gK, wave, lane, 0);
            MG_UNIT_STAMP(30, 1);
            MG_LITE(4);
            MG_LITE_DUMP;
        }
        const int64_t my_tiles = ((int64_t)blockIdx.x + 1) * a.n_tiles / gridDim.x - (int64_t)blockIdx.x * a.n_tiles / gridDim.x;
        if (my_tiles > 2) {
            mg_wait_producers(gprog + 24, 1);   // gfin[0], gfin[1]: both term buffers are free again
            mg_fused_gmm_terms<KK, LAT_F64>(gprog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, wave, lane, 1);
            if (wave < 2) mg_fused_gmm_finish(gprog, logp, a.B, a.n_tiles, gK, wave, lane, 1);
        }
    }
}


// -----------------------------------------------------------------------------------------
// launch
// -----------------------------------------------------------------------------------------
template <int KK, bool LAT_F64, bool FUSE, bool SPLIT>
static int mg_launch_cs_inst(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a,
                                 int buf_bytes, int lds, int grid, const mg_launch_events &ev) {
    // hipExtLaunchKernelGGL with NULL events is hipLaunchKernelGGL; with events the dispatch records its own begin and end
    hipExtLaunchKernelGGL((mg_frames_cs_kernel<KK, LAT_F64, FUSE, SPLIT>), dim3(grid), dim3(MG_CS_BLOCK), lds, p->ctx->stream, ev.start, ev.stop, 0,
                          (const float *)p->d_Epack, (const float *)p->d_mean32, (const double *)p->d_Erpack, (const double *)p->d_meanroot, lat,
                          (const int32_t *)g->d_i0, (const float4 *)g->d_w32, (const double *)g->d_wtap, (const float4 *)g->d_rootm,
                          (const mg_chunk *)g->d_chunks, out,
                          (const double *)p->d_gPpack, (const double *)p->d_gmPpad, (const double *)p->d_gconst, logp, a, (int)p->K,
                          (int)((p->L + 15) / 16), buf_bytes);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

template <int KK>
static int mg_launch_cs_kk(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a,
                               bool lat_f64, bool split, int buf_bytes, int lds, int grid, const mg_launch_events &ev) {
    if (logp) {
        // fused instances exist for <= 40 components: beyond that the mixture fragments no longer fit the
        // register budget next to the sweep (mg_frames_can_fuse_gmm refuses, so this is never reached)
        if constexpr (KK <= MG_FUSE_MAX_KK) {
            if (split)
                return lat_f64 ? mg_launch_cs_inst<KK, true, true, true>(p, g, lat, out, logp, a, buf_bytes, lds, grid, ev)
                               : mg_launch_cs_inst<KK, false, true, true>(p, g, lat, out, logp, a, buf_bytes, lds, grid, ev);
            return lat_f64 ? mg_launch_cs_inst<KK, true, true, false>(p, g, lat, out, logp, a, buf_bytes, lds, grid, ev)
                           : mg_launch_cs_inst<KK, false, true, false>(p, g, lat, out, logp, a, buf_bytes, lds, grid, ev);
        }
        mg_set_error("mg_step_frames_and_logp: no fused kernel for %d components", p->L);
        return MG_ERR_UNSUPPORTED;
    }
    if (split)
        return lat_f64 ? mg_launch_cs_inst<KK, true, false, true>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, ev)
                       : mg_launch_cs_inst<KK, false, false, true>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, ev);
    return lat_f64 ? mg_launch_cs_inst<KK, true, false, false>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, ev)
                   : mg_launch_cs_inst<KK, false, false, false>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, ev);
}

int mg_launch_frames_cs(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a, bool lat_f64,
                        bool split, int buf_bytes, int lds, int grid, const mg_launch_events &ev) {
    switch (p->KK) {
#ifndef MG_ONLY_KK10
        case 2: return mg_launch_cs_kk<2>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 4: return mg_launch_cs_kk<4>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 6: return mg_launch_cs_kk<6>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 8: return mg_launch_cs_kk<8>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
#endif
        case 10: return mg_launch_cs_kk<10>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
#ifndef MG_ONLY_KK10
        case 12: return mg_launch_cs_kk<12>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 14: return mg_launch_cs_kk<14>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 16: return mg_launch_cs_kk<16>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
#endif
        default: mg_set_error("mg_back_project_frames: MFMA path needs n_components <= 64"); return MG_ERR_UNSUPPORTED;
    }
}

template <int KK>
static int mg_cs_attr_kk() {
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if constexpr (KK <= MG_FUSE_MAX_KK) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_cs_kernel<KK, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    return MG_OK;
}
int mg_frames_cs_attributes() {
    int rc;
#ifndef MG_ONLY_KK10
    if ((rc = mg_cs_attr_kk<2>()) != MG_OK) return rc;
    if ((rc = mg_cs_attr_kk<4>()) != MG_OK) return rc;
    if ((rc = mg_cs_attr_kk<6>()) != MG_OK) return rc;
    if ((rc = mg_cs_attr_kk<8>()) != MG_OK) return rc;
#endif
    if ((rc = mg_cs_attr_kk<10>()) != MG_OK) return rc;
#ifndef MG_ONLY_KK10
    if ((rc = mg_cs_attr_kk<12>()) != MG_OK) return rc;
    if ((rc = mg_cs_attr_kk<14>()) != MG_OK) return rc;
    if ((rc = mg_cs_attr_kk<16>()) != MG_OK) return rc;
#endif
    return MG_OK;
}

// One launch per planner step (gfx950 / MI355X).
//
// GraphWalkPlanner evaluates every outgoing option of a node the same way (reference
// morphablegraphs/motion_generator/graph_walk_planner.py:184-226): draw n candidates from the option's mixture, score
// them against the step's constraints, keep the first minimum.  Option by option that is three launch-latency-bound
// kernels each (sampler, scorer, argmin + winner copy): 48 launches for a 16-option step, the GPU idle most of the time.
// mg_options_fused_kernel does the whole step in ONE launch:
//   * the workgroups are dealt over the options (option k owns workgroups wg0[k] .. wg0[k+1]); a wave owns one 16-row tile
//     of ONE mixture component, exactly the tiles of mg_gmm_sample_mfma_kernel (same Philox counters, same f64 MFMA chain:
//     the candidates are the same bits), keeps the tile in LDS and scores it right there with the arithmetic of
//     mg_score_mfma_kernel (channels by f64 MFMA with C-in = bias, residuals by mg_constraint_residual, summed in
//     constraint order: the errors are the same bits);
//   * first minimum: lanes -> wave (shuffles) -> workgroup (LDS) -> one partial {value, index} per workgroup in global
//     memory; the workgroup whose arrival is the last of its option (agent-scope release / counter / acquire) reduces the
//     option's partials, writes {index, error} and copies the winning latent behind them.  (value, index) pairs are
//     combined with a total order -- smaller value, then smaller index, NaN and +inf never win -- so the order of the
//     combination does not matter and the result is the one mg_argmin_gather_kernel finds.
// Per-step values (seeds, rows per component) travel in the kernel arguments; the per-option constants (fragment and
// constraint tables, buffers) sit in a small device table that is rewritten only when an option's pointers change.
#include <algorithm>
#include <cstring>

#include <hip/hip_ext.h>

#include "mg_internal.h"
#include "mg_gmm_device.h"
#include "mg_score_device.h"

#define MG_FUSED_MAX_OPTIONS 24
#ifndef MG_FUSED_WAVES_PER_SIMD
#define MG_FUSED_WAVES_PER_SIMD 3   // 167 VGPRs, no spills: 3 workgroups per CU (2: 44.5 us per 16 x 4096 step, 3: 40.9, 4: 51.3 with 36 spilled)
#endif

struct mg_fused_static {          // one option: what stays the same from step to step
    const double *cpack;          // sampler: [K][JT][KKg][64] fragments of chol^T
    const double *meanpad;        //          [K][JT*16]
    const double *Wpack;          // scorer:  [RT][KK][64]
    const double *bpad;           //          [RT*16]
    mg_score_args sa;             // constraint tables; sa.out = errors (n) float64, sa.B = n, sa.ld = ld of x
    void *x;                      // (n, ld) candidates, written
    void *result;                 // {int64 index, float64 error, float64 latent[Lg]}
    void *result_host;            // the same record in pinned host memory (or NULL): written by the kernel, so that a step that
                                  // wants its results on the host needs no copy after the launch, only the synchronisation
    double cumw[MG_SAMPLE_ARG_K]; // cumulative normalised mixture weights (float64): the device draw of the component counts
    int32_t K, Lg, KKg, KK, JT, RT;
};

struct mg_fused_dyn {             // one step: what changes every time (kernel argument)
    uint64_t seed[MG_FUSED_MAX_OPTIONS];
    int32_t counts[MG_FUSED_MAX_OPTIONS][MG_SAMPLE_ARG_K];
    int32_t wg0[MG_FUSED_MAX_OPTIONS + 1];
    // a rank's share of a sharded step (mg_options_step_rows): global rows [row_lo, row_hi) of every option's draw, held by
    // tiles [tile0[k], tile_end[k]); x and errors are indexed by (row - row_lo), the winner's index is global
    int32_t tile0[MG_FUSED_MAX_OPTIONS], tile_end[MG_FUSED_MAX_OPTIONS];
    int64_t row_lo, row_hi;
    // results on the host without a copy: an option's last workgroup writes its record into pinned memory and then, behind a
    // system-scope fence, the step's sequence number into the option's flag word there (NULL: no host copy wanted)
    unsigned long long *flags_host;
    unsigned long long seq;
    int32_t fenced;               // 1: the release / acquire form of the publication protocol (MG_OPT_OPTIONS_STEP 3; below)
};

struct mg_fused_devcounts {       // what mg_options_counts_kernel leaves for the step's main kernel (device memory)
    int32_t counts[MG_FUSED_MAX_OPTIONS][MG_SAMPLE_ARG_K];
    int32_t tile_end[MG_FUSED_MAX_OPTIONS];
};

// A workgroup's first minimum AND that candidate's latent vector: published with agent-scope stores (write-through), read by the
// option's last workgroup with agent-scope loads -- no release / acquire fence anywhere (an agent-scope release writes back the
// XCD's whole L2, into which the step has just written 10 MB of candidates: 1068 of them per step were the kernel's latency chain)
//
// THE PUBLICATION PROTOCOL'S CONTRACT.  What the fence-free form relies on is outside the HIP / LLVM memory model (relaxed atomics
// order nothing by themselves); it is the behaviour of gfx950 as the AMDGPU backend's memory-model table for gfx942 / gfx950 states
// it and as this image's toolchain compiles it -- verified on ROCm 7.2.0 (HIP 7.2, AMD clang 22), MI355X, by the parity tests, by
// tests/test_gpu_adaptors.py::test_planner_steps_publish_complete_records_fenced_or_not (500 steps, every record checked against
// the device buffers, the fenced form beside it) and by tools/probes/options_step_soak.py (3000 steps):
//   (1) a relaxed atomic store at AGENT scope compiles to global_store ... sc1: it writes through the XCD's L2 to the level the
//       agent's XCDs share (memory / MALL), it does not linger in the non-coherent L2 of the writing XCD;
//       at SYSTEM scope (the pinned host record, the flag) to global_store ... sc0 sc1: through to the fabric;
//   (2) vmcnt counts a store down when its write has been ACKNOWLEDGED by that level, not when it has left the CU: after
//       s_waitcnt vmcnt(0) the wave's write-through stores are visible to every XCD (resp. the host);
//   (3) a relaxed atomic load at agent scope compiles to global_load ... sc1: it misses the reading XCD's L2 for such lines and
//       reads the shared level -- it cannot return a stale copy cached before the store;
//   (4) the counter is a returning atomic add performed AT the shared level (device-scope atomics execute in the memory-side
//       atomics unit): its order against (2) is "all of this workgroup's partial stores acknowledged, barrier, then the add";
//       the last arrival's loads are issued after its add has returned.
// So: partial stores acknowledged (2) -> barrier -> add (4) ... last add returns -> loads (3) see (1).  The host record the same
// way at system scope: record stores acknowledged, barrier, flag store.  A compiler that stops mapping scopes to sc0 / sc1 this
// way, or firmware that acknowledges write-through stores early, breaks it silently -- hence the test, and hence
// MG_OPT_OPTIONS_STEP = 3, which runs the SAME kernel with a release fence before the add, an acquire fence after it and a
// system-scope release before the flag (the memory model's own protocol; ~8 us slower per 16 x 4096 step, same records).
struct mg_fused_partial { double v; int64_t i; double row[4 * MG_MAX_KK]; };

// The component counts of a step drawn ON THE DEVICE (mg_options_step_device_counts): a multinomial(n, weights) draw per option
// as the histogram of n categorical draws, keyed by the option's seed --
//     u_i = (Philox4x32-10(counter = (i >> 2, 0, 0, MG_COUNTS_TAG), key = seed)[i & 3] + 0.5) / 2^32,   i = 0 .. n-1,
//     counts[c] = #{i : cum[c-1] <= u_i < cum[c]},  cum = cumulative normalised weights in float64, the last component takes the rest
// -- distributed like numpy.random.multinomial's counts (what GaussianMixture.sample draws, reference
// motion_primitive.py:182-189), not NumPy's stream: the same status as the device sampler.  One workgroup per option writes the
// step's mg_fused_dyn into device memory, where mg_options_fused_kernel<., true> reads it; the host, not knowing the counts,
// sizes every option's share of the grid for the most tiles n candidates in K components can take (n / 16 + K).
#define MG_COUNTS_TAG 0x636e7473u
struct mg_counts_args {
    uint64_t seed[MG_FUSED_MAX_OPTIONS];
    int64_t n;
};
// the draw of one option's counts by a 256-thread workgroup (every thread of it must call)
__device__ __forceinline__ void mg_counts_draw(const mg_fused_static &o, const uint64_t seed, const int64_t n, mg_fused_devcounts *__restrict__ dc, const int k,
                                               int (*below)[MG_SAMPLE_ARG_K]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = o.K;
    double cum[MG_SAMPLE_ARG_K];     // the thresholds in registers: never-below from K - 1 on, so that the comparisons below need no bound
#pragma unroll
    for (int c = 0; c < MG_SAMPLE_ARG_K; c++) cum[c] = o.cumw[c];
#pragma unroll
    for (int c = 0; c < MG_SAMPLE_ARG_K; c++) cum[c] = c < K - 1 ? cum[c] : -1.0;   // (u >= 0: never below)
    int lt[MG_SAMPLE_ARG_K];   // this thread's draws below cum[c]
#pragma unroll
    for (int c = 0; c < MG_SAMPLE_ARG_K; c++) lt[c] = 0;
    const int64_t groups = (n + 3) >> 2;
    for (int64_t j = tid; j < groups; j += 256) {
        uint32_t rr[4];
        mg_philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), 0u, MG_COUNTS_TAG, (uint32_t)seed, (uint32_t)(seed >> 32), rr);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double u = 4 * j + e < n ? ((double)rr[e] + 0.5) * (1.0 / 4294967296.0) : 2.0;   // (beyond the draw: below nothing)
#pragma unroll
            for (int c = 0; c < MG_SAMPLE_ARG_K; c++) lt[c] += u < cum[c] ? 1 : 0;
        }
    }
#pragma unroll
    for (int c = 0; c < MG_SAMPLE_ARG_K; c++) {
        int v = lt[c];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) below[wave][c] = v;
    }
    __syncthreads();
    if (tid < MG_SAMPLE_ARG_K) {     // lane c: component c's count
        const int c = tid;
        auto cum_at = [&](int q) { return q < 0 ? 0 : (q < K - 1 ? below[0][q] + below[1][q] + below[2][q] + below[3][q] : (int)n); };
        const int cnt = c < K ? cum_at(c) - cum_at(c - 1) : 0;
        dc->counts[k][c] = cnt;
        int tiles = (cnt + 15) / 16;
        for (int off = 8; off > 0; off >>= 1) tiles += __shfl_down(tiles, off, 16);
        if (c == 0) dc->tile_end[k] = tiles;
    }
    __syncthreads();                 // (below[] may be used again)
}
// ... as a launch of its own in front of the step's kernel (9.3 us, nearly all of it launch and drain: the draw is 4096 Philox
// values per option), when the counts were not drawn by the step before (see mg_options_fused_kernel's `next`)
__global__ __launch_bounds__(256) void mg_options_counts_kernel(const mg_fused_static *__restrict__ tab, const mg_counts_args a,
                                                               mg_fused_devcounts *__restrict__ dc) {
    __shared__ int below[4][MG_SAMPLE_ARG_K];
    mg_counts_draw(tab[blockIdx.x], a.seed[blockIdx.x], a.n, dc, blockIdx.x, below);
}

#ifndef MG_OPT_PAIRED_FRAGMENTS
#define MG_OPT_PAIRED_FRAGMENTS 1
#endif
template <bool X_F64, bool DYN_DEV>
__global__ __launch_bounds__(256, MG_FUSED_WAVES_PER_SIMD) void mg_options_fused_kernel(const mg_fused_static *__restrict__ tab, const mg_fused_dyn dyn_arg,
                                                              const mg_fused_devcounts *__restrict__ dyn_dev,
                                                              mg_fused_devcounts *__restrict__ next_dev, int32_t *__restrict__ counts_host,
                                                              const int n_options, const int wave_doubles,
                                                              mg_fused_partial *__restrict__ partials, int32_t *__restrict__ counters) {
    const mg_fused_dyn &dyn = dyn_arg;   // the step's values; DYN_DEV: the component counts and the tile count come from dyn_dev
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    __shared__ int s_last;
    __shared__ int s_below[4][MG_SAMPLE_ARG_K];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;
    const int wg = blockIdx.x;
    int k = 0;
    while (k + 1 < n_options && wg >= dyn.wg0[k + 1]) k++;
    const mg_fused_static &o = tab[k];
    const int K = o.K, Lg = o.Lg, KKg = o.KKg, KK = o.KK, JT = o.JT, RT = o.RT, L = o.sa.L, n = o.sa.n;
    const int64_t ld = o.sa.ld;
    const int ZS = 4 * KKg + 1, vs = RT * 16 + 1;
    double *zt = (double *)smem + (size_t)wave * wave_doubles;   // [16][ZS] standard normals, then the candidates of the tile
    double *vals = zt + 16 * ZS;                                  // [16][vs] pose channels of the tile
    double *resid = vals + 16 * vs;                               // [n][16]

    // this wave's tile: rows [row0, row0 + nrow) of component c (rows grouped by component, tiles never straddle two)
    const int64_t t = dyn.tile0[k] + (int64_t)(wg - dyn.wg0[k]) * 4 + wave;
    const int64_t row_lo = dyn.row_lo, row_hi = dyn.row_hi;
    int c = 0;
    int64_t row_c = 0, tile_c = 0, rows_next, tiles_next;
    int32_t cnts[MG_SAMPLE_ARG_K];   // (one 64-byte line: a single trip to memory where the counts were drawn on the device)
#pragma unroll
    for (int q = 0; q < MG_SAMPLE_ARG_K; q++) cnts[q] = DYN_DEV ? dyn_dev->counts[k][q] : dyn.counts[k][q];
    const int tile_end_k = DYN_DEV ? dyn_dev->tile_end[k] : dyn.tile_end[k];
    if (DYN_DEV && counts_host && wg == dyn.wg0[k] && tid < MG_SAMPLE_ARG_K) {
        // the counts this step is drawn with, for the host: one 64-byte system-scope store, long before the step's flags
        int mine = cnts[0];
#pragma unroll
        for (int q = 1; q < MG_SAMPLE_ARG_K; q++) mine = tid == q ? cnts[q] : mine;
        __hip_atomic_store(counts_host + k * MG_SAMPLE_ARG_K + tid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (;;) {
        int cnt = cnts[0];
#pragma unroll
        for (int q = 1; q < MG_SAMPLE_ARG_K; q++) cnt = c == q ? cnts[q] : cnt;
        rows_next = row_c + cnt;
        tiles_next = tile_c + (cnt + 15) / 16;
        if (c + 1 < K && t >= tiles_next) { c++; row_c = rows_next; tile_c = tiles_next; }
        else break;
    }
    const bool active = t < tiles_next && t < tile_end_k;
    double best = INFINITY;
    int64_t bi = INT64_MAX;
    if (active) {
        const int64_t row0 = row_c + (t - tile_c) * 16;
        const int nrow = (int)((rows_next - row0) < 16 ? (rows_next - row0) : 16);
        const uint64_t seed = dyn.seed[k];
        for (int e = lane; e < 16 * KKg; e += 64) {
            const int r = e & 15, q = e >> 4;
            const int64_t b = row0 + r;
            double z4[4];
            mg_normal4(b, q, seed, z4);
#pragma unroll
            for (int j = 0; j < 4; j++) zt[r * ZS + 4 * q + j] = z4[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double za[MG_MAX_KK];
#pragma unroll
        for (int kk = 0; kk < MG_MAX_KK; kk++) za[kk] = (kk < KKg && 4 * kk + g < Lg) ? zt[cl * ZS + 4 * kk + g] : 0.0;
        // x = mu + z L^T; the tile goes to memory in the caller's type and stays in LDS, rounded the same way, for the scorer
#pragma unroll
        for (int it = 0; it < MG_MAX_KK / 4; it++) {
            if (it < JT) {
                const double m = o.meanpad[((size_t)c * JT + it) * 16 + cl];
                mg_f64x4 acc = {m, m, m, m};
#if MG_OPT_PAIRED_FRAGMENTS
                // (the paired copy behind the image, mg_host.hip: one 16-byte load per two k-steps; KKg is even, 4 (it + 1) a multiple of four)
                typedef double mg_f64x2 __attribute__((ext_vector_type(2)));
                const mg_f64x2 *cp2 = (const mg_f64x2 *)(o.cpack + (size_t)K * JT * KKg * 64) + (((size_t)c * JT + it) * (KKg / 2)) * 64 + lane;
#pragma unroll
                for (int q = 0; q < MG_MAX_KK / 2; q++)
                    if (2 * q < KKg && 2 * q < 4 * (it + 1)) {
                        const mg_f64x2 v = cp2[q * 64];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(za[2 * q], v[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(za[2 * q + 1], v[1], acc, 0, 0, 0);
                    }
#else
                const double *cp = o.cpack + (((size_t)c * JT + it) * KKg) * 64 + lane;
#pragma unroll
                for (int kk = 0; kk < MG_MAX_KK; kk++)
                    if (kk < KKg && kk < 4 * (it + 1)) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(za[kk], cp[kk * 64], acc, 0, 0, 0);
#endif
                const int i = 16 * it + cl;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = g + 4 * r;
                    if (i < Lg) {
                        const double xv = X_F64 ? acc[r] : (double)(float)acc[r];
                        zt[row * ZS + i] = xv;
                        const int64_t gr = row0 + row;
                        if (row < nrow && gr >= row_lo && gr < row_hi) {
                            if (X_F64) ((double *)o.x)[(gr - row_lo) * ld + i] = acc[r];
                            else ((float *)o.x)[(gr - row_lo) * ld + i] = (float)acc[r];
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // pose channels of the tile: X (16 x L) . W^T + bias on the f64 matrix pipe
        double xf[MG_MAX_KK];
#pragma unroll
        for (int kk = 0; kk < MG_MAX_KK; kk++) xf[kk] = (kk < KK && 4 * kk + g < L && cl < nrow) ? zt[cl * ZS + 4 * kk + g] : 0.0;
        for (int rt = 0; rt < RT; rt++) {
            const double *wp = o.Wpack + ((size_t)rt * KK) * 64 + lane;
            const double c0 = o.bpad[rt * 16 + cl];
            mg_f64x4 acc = {c0, c0, c0, c0};
#pragma unroll
            for (int kk = 0; kk < MG_MAX_KK; kk++)
                if (kk < KK) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[kk], wp[kk * 64], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) vals[(g + 4 * r) * vs + rt * 16 + cl] = acc[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int e = lane; e < 16 * n; e += 64) {
            const int cand = e & 15, cc = e >> 4;
            const double *v = vals + cand * vs;
            resid[cc * 16 + cand] = mg_constraint_residual(o.sa, cc, [&](int row) { return v[row]; });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < nrow && row0 + lane >= row_lo && row0 + lane < row_hi) {
            double err = 0.0;
            for (int cc = 0; cc < n; cc++) err += resid[cc * 16 + lane];
            ((double *)o.sa.out)[row0 + lane - row_lo] = err;
            if (err < INFINITY) { best = err; bi = row0 + lane; }   // NaN and +inf never win
        }
    }
    // first minimum of the workgroup
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const long long oi = __shfl_down((long long)bi, off, 64);
        mg_min_combine(best, bi, ov, (int64_t)oi);
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    const int nwg = dyn.wg0[k + 1] - dyn.wg0[k];
    // the workgroup's first minimum (every thread: four combines from LDS); the wave whose tile holds that row publishes its latent
    // vector from the tile it still has in LDS (a workgroup without a finite error publishes the block's first row if it holds it: what
    // a step without any finite error returns)
    best = sv[0]; bi = si[0];
    for (int w = 1; w < 4; w++) mg_min_combine(best, bi, sv[w], si[w]);
    {
        const int64_t pub = bi != INT64_MAX ? bi : row_lo;
        const int64_t row0 = row_c + (t - tile_c) * 16;
        const int64_t row_end = rows_next < row0 + 16 ? rows_next : row0 + 16;   // (a component's last tile may be short: the next rows are another tile's)
        if (active && pub >= row0 && pub < row_end && lane < Lg)
            __hip_atomic_store(&partials[wg].row[lane], zt[(int)(pub - row0) * ZS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        __hip_atomic_store(&partials[wg].v, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partials[wg].i, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (also this wave's candidates and errors; the write-through stores above are complete)
    if (dyn.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(&counters[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == nwg - 1) ? 1 : 0;
    }
    __syncthreads();
    if (dyn.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (DYN_DEV && next_dev != nullptr && wg == dyn.wg0[k]) {
        // The option's first workgroup, its own tile published, draws the counts the NEXT step will want if its seed is this one's + 1
        // (a planner counts its steps): that step then starts without a counts kernel in front (8.5 us + the gap between two launches).
        // Off the critical path -- the first workgroup is done long before the option's last -- and lost if the next step asks otherwise.
        mg_counts_draw(o, dyn.seed[k] + 1, o.sa.B, next_dev, k, s_below);
    }
    if (!s_last) return;
    // the option's last workgroup: every other one has published its partial
    best = INFINITY; bi = INT64_MAX;
    for (int w = tid; w < nwg; w += 256) {
        const mg_fused_partial *pp = &partials[dyn.wg0[k] + w];
        const double pv = __hip_atomic_load(&pp->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t pi = __hip_atomic_load(&pp->i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mg_min_combine(best, bi, pv, pi);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const long long oi = __shfl_down((long long)bi, off, 64);
        mg_min_combine(best, bi, ov, (int64_t)oi);
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) mg_min_combine(best, bi, sv[w], si[w]);
        si[1] = bi;                                  // the winner as the workgroups published it (INT64_MAX: none was finite)
        if (bi == INT64_MAX || bi < row_lo || bi >= row_hi) { bi = row_lo; best = INFINITY; }   // (an index outside the block cannot happen; never gather out of bounds)
        ((int64_t *)o.result)[0] = bi;
        ((double *)o.result)[1] = best;
        if (o.result_host) {    // (system-scope stores: straight to the pinned record, complete when vmcnt says so -- no fence below)
            __hip_atomic_store((int64_t *)o.result_host, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store((double *)o.result_host + 1, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        si[0] = bi;
        counters[k] = 0;   // ready for the next launch (stream ordered)
    }
    __syncthreads();
    // whose row: the workgroup that published the winner's index (no finite error anywhere: the first workgroup, which then published
    // the block's first row)
    const int64_t widx = si[1];
    if (tid == 0) s_last = 0;
    __syncthreads();
    if (widx != INT64_MAX)
        for (int w = tid; w < nwg; w += 256)
            if (__hip_atomic_load(&partials[dyn.wg0[k] + w].i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == widx) s_last = w;
    __syncthreads();
    const mg_fused_partial *win = &partials[dyn.wg0[k] + s_last];
    double *row = (double *)((char *)o.result + 16);
    double *row_h = o.result_host ? (double *)((char *)o.result_host + 16) : nullptr;
    for (int i = tid; i < Lg; i += 256) {
        const double v = __hip_atomic_load(&win->row[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        row[i] = v;
        if (row_h) __hip_atomic_store(row_h + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (dyn.flags_host) {   // the record is complete on the host before its flag says so: every thread's record stores have been
        // acknowledged (vmcnt) before the barrier, the flag leaves after it -- a system-scope release fence here would write back
        // the whole L2 (the step's candidates) first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dyn.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope
        __syncthreads();
        if (tid == 0) __hip_atomic_store(dyn.flags_host + k, dyn.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int mg_options_fused_attributes() {
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_options_fused_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_options_fused_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_options_fused_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_options_fused_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    return MG_OK;
}

// Can this step run as one launch?  Every option on the MFMA sampler (prefix sums in the arguments) and the MFMA scorer.
bool mg_options_can_fuse(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n) {
    if (n_options < 1 || n_options > MG_FUSED_MAX_OPTIONS || n < 1 || n > (int64_t)1 << 30) return false;
    mg_context *ctx = prims[0]->ctx;
    if (ctx->opt[MG_OPT_FORCE_VALU_SAMPLE] || ctx->opt[MG_OPT_FORCE_VALU_SCORE] || ctx->opt[MG_OPT_OPTIONS_STEP] == 1) return false;
    for (int k = 0; k < n_options; k++) {
        const mg_primitive *p = prims[k];
        const mg_constraint_set *cs = csets[k];
        if (!mg_gmm_sample_takes_host_prefix(p) || p->KK <= 0 || !cs || cs->prim != p || !cs->d_Wpack) return false;
        const int64_t wd = 16 * (4 * p->KKg + 1) + 16 * (cs->RT * 16 + 1) + 16 * (int64_t)std::max(cs->n, 1);
        if (4 * wd * 8 > 150 * 1024) return false;
    }
    return true;
}

// counts == NULL: the component counts are drawn on the device (mg_options_counts_kernel; whole draws only: row_begin = 0,
// row_count = n).  records_host / counts_host (device-visible pinned memory, or NULL): where the kernels also leave the result
// records (record k at k * result_stride) and the drawn counts [n_options][MG_SAMPLE_ARG_K].
int mg_launch_options_fused(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n,
                            const int64_t *const *counts, const uint64_t *seeds, void *const *x_dev, int xdt, const int64_t *ld,
                            double *const *errors_dev, void *results_dev, int64_t result_stride, int64_t row_begin, int64_t row_count,
                            void *records_host, int32_t *counts_host, unsigned long long *flags_host, unsigned long long seq) {
    mg_context *ctx = prims[0]->ctx;
    const bool dev_counts = counts == nullptr;
    if (dev_counts && (row_begin != 0 || row_count != n)) {
        mg_set_error("mg_options_step_device_counts: the device draws the counts of whole draws only");
        return MG_ERR_UNSUPPORTED;
    }
    mg_counts_args ca;
    memset(&ca, 0, sizeof(ca));
    ca.n = n;
    std::vector<mg_fused_static> tab((size_t)n_options);
    mg_fused_dyn dyn;
    memset(&dyn, 0, sizeof(dyn));
    memset(tab.data(), 0, tab.size() * sizeof(mg_fused_static));
    int wave_doubles = 0;
    for (int k = 0; k < n_options; k++) {
        mg_primitive *p = prims[k];
        const mg_constraint_set *cs = csets[k];
        mg_fused_static &o = tab[k];
        o.cpack = p->d_gcholpack; o.meanpad = p->d_gmeanpad; o.Wpack = cs->d_Wpack; o.bpad = cs->d_bpad;
        mg_score_args &a = o.sa;
        a.W = cs->d_W; a.bias = cs->d_bias; a.par = cs->d_par; a.woff = cs->d_woff; a.chain = cs->d_chain; a.choff = cs->d_choff;
        a.align = cs->d_align; a.align_cand = nullptr; a.pose = cs->d_pose; a.lat = x_dev[k]; a.out = errors_dev[k]; a.res = nullptr; a.B = n; a.ld = ld[k];
        a.n = cs->n; a.nch = cs->nch; a.L = p->L;
        o.x = x_dev[k]; o.result = (char *)results_dev + (size_t)k * result_stride;
        o.result_host = records_host ? (char *)records_host + (size_t)k * result_stride : nullptr;
        {
            double wsum = 0.0, acc = 0.0;
            for (int c = 0; c < p->K; c++) wsum += p->gw[c];
            for (int c = 0; c < p->K && c < MG_SAMPLE_ARG_K; c++) { acc += p->gw[c]; o.cumw[c] = acc / wsum; }
        }
        o.K = p->K; o.Lg = p->Lg; o.KKg = p->KKg; o.KK = p->KK; o.JT = (p->Lg + 15) / 16; o.RT = cs->RT;
        wave_doubles = std::max(wave_doubles, 16 * (4 * p->KKg + 1) + 16 * (cs->RT * 16 + 1) + 16 * std::max(cs->n, 1));
        int64_t tiles = 0, rows = 0, t_first = -1, t_last = -1;
        if (dev_counts) {   // the grid's share of this option: the most tiles n rows in K components can take
            ca.seed[k] = seeds[k];
            dyn.seed[k] = seeds[k];
            dyn.tile0[k] = 0;
            dyn.wg0[k + 1] = dyn.wg0[k] + (int32_t)((n / 16 + p->K + 3) / 4);
            continue;
        }
        for (int c = 0; c < p->K; c++) {
            if (counts[k][c] < 0 || counts[k][c] > n) { mg_set_error("mg_options_step: counts[%d][%d] out of range", k, c); return MG_ERR_INVALID_ARGUMENT; }
            dyn.counts[k][c] = (int32_t)counts[k][c];
            // the tiles that hold the first and the last row of the block
            if (t_first < 0 && row_begin < rows + counts[k][c]) t_first = tiles + (row_begin - rows) / 16;
            if (t_last < 0 && row_begin + row_count - 1 < rows + counts[k][c]) t_last = tiles + (row_begin + row_count - 1 - rows) / 16;
            rows += counts[k][c];
            tiles += (counts[k][c] + 15) / 16;
        }
        if (rows != n) { mg_set_error("mg_options_step: counts of option %d sum to %lld, expected %lld", k, (long long)rows, (long long)n); return MG_ERR_INVALID_ARGUMENT; }
        dyn.seed[k] = seeds[k];
        dyn.tile0[k] = (int32_t)t_first;
        dyn.tile_end[k] = (int32_t)(t_last + 1);
        dyn.wg0[k + 1] = dyn.wg0[k] + (int32_t)((t_last + 1 - t_first + 3) / 4);
    }
    dyn.row_lo = row_begin; dyn.row_hi = row_begin + row_count;
    dyn.flags_host = flags_host; dyn.seq = seq;
    dyn.fenced = ctx->opt[MG_OPT_OPTIONS_STEP] == 3 ? 1 : 0;
    const int total_wg = dyn.wg0[n_options];
    // the static table: uploaded when it differs from the one on the device (a planner reuses its buffers and sets, so: rarely)
    const size_t tab_bytes = tab.size() * sizeof(mg_fused_static);
    if (!ctx->fused_tab_dev || ctx->fused_tab_host.size() != tab_bytes || memcmp(ctx->fused_tab_host.data(), tab.data(), tab_bytes) != 0) {
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // no launch may still be reading the old table
        if (!ctx->fused_tab_dev) MG_HIP_CHECK(hipMalloc(&ctx->fused_tab_dev, MG_FUSED_MAX_OPTIONS * sizeof(mg_fused_static)));
        MG_HIP_CHECK(hipMemcpy(ctx->fused_tab_dev, tab.data(), tab_bytes, hipMemcpyHostToDevice));
        ctx->fused_tab_host.assign((const unsigned char *)tab.data(), (const unsigned char *)tab.data() + tab_bytes);
    }
    if (!ctx->fused_counters) {
        MG_HIP_CHECK(hipMalloc(&ctx->fused_counters, MG_FUSED_MAX_OPTIONS * sizeof(int32_t)));
        MG_HIP_CHECK(hipMemset(ctx->fused_counters, 0, MG_FUSED_MAX_OPTIONS * sizeof(int32_t)));
    }
    if (ctx->fused_partials_n < total_wg) {
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->fused_partials) MG_HIP_CHECK(hipFree(ctx->fused_partials));
        ctx->fused_partials = nullptr; ctx->fused_partials_n = 0;
        const int cap = std::max(total_wg, 2048);
        MG_HIP_CHECK(hipMalloc(&ctx->fused_partials, (size_t)cap * sizeof(mg_fused_partial)));
        ctx->fused_partials_n = cap;
    }
    const size_t lds = (size_t)4 * wave_doubles * 8;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool timed = mg_prof_kernel(ctx, 7, -1, &ev0, &ev1);
    const mg_fused_static *tabd = (const mg_fused_static *)ctx->fused_tab_dev;
    mg_fused_partial *part = (mg_fused_partial *)ctx->fused_partials;
    int32_t *ctr = (int32_t *)ctx->fused_counters;
    if (dev_counts) {
        if (!ctx->fused_dyn_dev) MG_HIP_CHECK(hipMalloc(&ctx->fused_dyn_dev, 2 * sizeof(mg_fused_devcounts)));
        // were this step's counts drawn by the step before?
        auto &fn = ctx->fused_next;
        bool hit = fn.valid && fn.n_options == n_options && fn.n == n && ctx->opt[MG_OPT_OPTIONS_STEP] != 2;
        for (int k = 0; k < n_options && hit; k++) hit = fn.seeds[k] == seeds[k] && fn.prims[k] == (const void *)prims[k];
        const int slot = hit ? fn.slot : 0;
        mg_fused_devcounts *dd = (mg_fused_devcounts *)ctx->fused_dyn_dev + slot, *dnext = (mg_fused_devcounts *)ctx->fused_dyn_dev + (slot ^ 1);
        if (!hit) hipLaunchKernelGGL(mg_options_counts_kernel, dim3(n_options), dim3(256), 0, ctx->stream, tabd, ca, dd);
        fn.valid = false; fn.n_options = n_options; fn.n = n; fn.slot = slot ^ 1;   // (valid once the launch below has been accepted)
        for (int k = 0; k < n_options; k++) { fn.seeds[k] = seeds[k] + 1; fn.prims[k] = (const void *)prims[k]; }
        if (xdt == MG_F64)
            hipExtLaunchKernelGGL((mg_options_fused_kernel<true, true>), dim3(total_wg), dim3(256), lds, ctx->stream, timed ? ev0 : nullptr, timed ? ev1 : nullptr, 0,
                                  tabd, dyn, (const mg_fused_devcounts *)dd, dnext, counts_host, (int)n_options, wave_doubles, part, ctr);
        else
            hipExtLaunchKernelGGL((mg_options_fused_kernel<false, true>), dim3(total_wg), dim3(256), lds, ctx->stream, timed ? ev0 : nullptr, timed ? ev1 : nullptr, 0,
                                  tabd, dyn, (const mg_fused_devcounts *)dd, dnext, counts_host, (int)n_options, wave_doubles, part, ctr);
    } else if (xdt == MG_F64) {
        hipExtLaunchKernelGGL((mg_options_fused_kernel<true, false>), dim3(total_wg), dim3(256), lds, ctx->stream, timed ? ev0 : nullptr, timed ? ev1 : nullptr, 0,
                              tabd, dyn, (const mg_fused_devcounts *)nullptr, (mg_fused_devcounts *)nullptr, (int32_t *)nullptr, (int)n_options, wave_doubles, part, ctr);
    } else {
        hipExtLaunchKernelGGL((mg_options_fused_kernel<false, false>), dim3(total_wg), dim3(256), lds, ctx->stream, timed ? ev0 : nullptr, timed ? ev1 : nullptr, 0,
                              tabd, dyn, (const mg_fused_devcounts *)nullptr, (mg_fused_devcounts *)nullptr, (int32_t *)nullptr, (int)n_options, wave_doubles, part, ctr);
    }
    MG_HIP_CHECK(hipGetLastError());
    if (dev_counts) ctx->fused_next.valid = true;
    return MG_OK;
}

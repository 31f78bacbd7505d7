// Placement of large kernel outputs (gfx950 / MI355X).
//
// The frames kernel writes B x 49 KB as thousands of concurrent streams of ~1 KB pieces.  Stand-alone that store
// stream takes 63 us for 404 MB in some parts of the card's memory (the rate of a plain fill) and 79 us in others;
// the frames kernel itself 85 vs 93-95 us.  What the class depends on was narrowed down with tools/chan_probe*.hip:
//   * not the base offset inside physically contiguous memory (one 1 GiB virtual-memory chunk: 78-81 us at every
//     2 MiB offset, and at +256 B ... +1 MiB), not the stride between the streams, not the order in which
//     workgroups walk their units, not a time skew between workgroups, not cache carry-over between launches
//     (two buffers written alternately keep their classes);
//   * the physical region: of 160 buffers of 404 MB held at once, numbers 2-5, 28, 117, 126-138 and 141-159 were
//     fast -- long runs of neighbours, i.e. multi-gigabyte regions of one class;
//   * a plain fill runs at the same rate everywhere (59-61 us), so no simple kernel change can see or avoid it.
// So the library probes for placement: mg_device_malloc_placed times the store pattern against a fill on each
// candidate allocation and keeps the first fast one, holding the rejected candidates until then so that the next
// one comes from other memory.
#include <algorithm>
#include <map>
#include <vector>
#include <hip/hip_ext.h>

#include "mg_internal.h"

typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef f32x4p f32x4pu __attribute__((aligned(4)));

// the sweep's store stream for a dense (n_cand, 156, 79) float block: 8 storing waves per workgroup, a wave owns two
// of a tile's 16 candidates, one instruction stores three 316-byte rows (952 consecutive bytes), 4 chunks of 39 rows
#define MG_PP_T 156
#define MG_PP_D 79
#define MG_PP_NF 39
#define MG_PP_NCH 4
__global__ __launch_bounds__(512) void mg_placement_pattern_kernel(float *out, int ntiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fsub = lane / 20, ql = lane % 20;
    const bool on = lane < 60;
    const int U = ntiles * MG_PP_NCH, per = (U + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int s = 0; s < per; s++) {
        const int u = blockIdx.x * per + s;
        if (u >= U) break;
        const int tile = u / MG_PP_NCH, chunk = (u % MG_PP_NCH + blockIdx.x) % MG_PP_NCH;
        for (int f0 = 0; f0 < MG_PP_NF; f0 += 3)
            for (int half = 0; half < 2; half++) {
                const size_t cand = (size_t)tile * 16 + wave + 8 * half;
                const int f = f0 + fsub;
                if (on && f < MG_PP_NF) {
                    float *p = out + (cand * MG_PP_T + (size_t)(chunk * MG_PP_NF + f)) * MG_PP_D + (ql == 19 ? 75 : 4 * ql);
                    const f32x4pu v = {0.f, 0.f, 0.f, 0.f};
                    *(f32x4pu *)p = v;
                }
            }
    }
}
__global__ __launch_bounds__(256) void mg_placement_fill_kernel(f32x4p *buf, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const f32x4p v = {0.f, 0.f, 0.f, 0.f}; buf[i] = v; }
}

#define MG_PLACED_FAST_RATIO 1.15                  // measured: 1.04-1.06 in the fast class, 1.25-1.33 in the slow one
// ... and an absolute floor for probes that fill the chip (256 tiles and more): a process can meet memory where the plain fill is
// slow as well (fill 77 us, pattern 80 us for the bench's 404 MB: ratio 1.04, 5.0 TB/s -- against 61-63.5 us = 6.4-6.6 TB/s in
// the fast class and 54-57 us in the best regions), which the ratio alone waves through
#define MG_PLACED_FAST_TBPS 6.0
// Hysteresis: the scan STOPS at the first candidate of 6.0 TB/s and more; a region's CLASS -- what the frames kernels' choice
// and every report go by, decided once from the scan's own measurement and kept for the region's life -- is fast from 5.9 TB/s
// on.  (Round 3's bench met a buffer at 5.996: one threshold for both made a 0.07 % margin flip the kernel choice, and a second
// probe of the same buffer contradict the arena.)
#define MG_PLACED_CLASS_TBPS 5.9
#define MG_PLACED_DEEP_CANDIDATES 400       // the default scan's last resort (see mg_region_create)
static double mg_placement_tbps(int64_t bytes, double pattern_us) {   // 0: the probe is too small to fill the chip (or was not timed)
    const int64_t cand_bytes = (int64_t)MG_PP_T * MG_PP_D * 4;
    const int64_t ntiles = std::min<int64_t>(bytes / (16 * cand_bytes), 1 << 20);
    if (ntiles < 256 || pattern_us <= 0.0) return 0.0;
    return (double)(ntiles * 16 * cand_bytes) / pattern_us * 1e-6;
}
static bool mg_placement_is_fast(int64_t bytes, double ratio, double pattern_us, double fast_ratio, double floor_tbps = MG_PLACED_FAST_TBPS) {
    if (ratio > fast_ratio) return false;
    const double tbps = mg_placement_tbps(bytes, pattern_us);
    return tbps == 0.0 || tbps >= floor_tbps;
}

int mg_probe_placement(mg_context *ctx, void *buf, int64_t bytes, double *ratio, double *pattern_us) {
    const int64_t cand_bytes = (int64_t)MG_PP_T * MG_PP_D * 4;
    const int ntiles = (int)std::min<int64_t>(bytes / (16 * cand_bytes), 1 << 20);
    if (ntiles < 64) { *ratio = 1.0; *pattern_us = 0.0; return MG_OK; }
    const size_t n4 = (size_t)ntiles * 16 * (size_t)cand_bytes / 16;
    const int grid = std::min(ntiles * MG_PP_NCH, std::max(1, ctx->n_cu));
    // Every launch carries its own start / stop events (attached to the dispatch: the packet's own timestamps), and the kernels'
    // durations are summed -- not the time between two markers around a row of launches: under a tracing profiler every launch gap
    // grows by microseconds, a marker-to-marker measurement then calls every buffer slow and the arena picks the other kernel.
    hipStream_t st = ctx->stream;
    const int warm = 2, reps = 6;
    hipEvent_t ev[4 * reps];
    for (auto &e : ev) MG_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    auto pattern = [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(mg_placement_pattern_kernel, dim3(grid), dim3(512), 0, st, a, b, 0, (float *)buf, ntiles); };
    auto fill = [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(mg_placement_fill_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a, b, 0, (f32x4p *)buf, n4); };
    for (int i = 0; i < warm; i++) { fill(nullptr, nullptr); pattern(nullptr, nullptr); }
    for (int i = 0; i < reps; i++) fill(ev[2 * i], ev[2 * i + 1]);
    for (int i = 0; i < reps; i++) pattern(ev[2 * reps + 2 * i], ev[2 * reps + 2 * i + 1]);
    hipError_t e = hipStreamSynchronize(st);
    float ms_fill = 0.f, ms_pat = 0.f;
    for (int i = 0; i < reps && e == hipSuccess; i++) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        ms_fill += ms;
        if (e == hipSuccess) { e = hipEventElapsedTime(&ms, ev[2 * reps + 2 * i], ev[2 * reps + 2 * i + 1]); ms_pat += ms; }
    }
    if (e == hipSuccess) e = hipGetLastError();
    for (auto &x : ev) (void)hipEventDestroy(x);
    if (e != hipSuccess) return mg_hip_fail(e, "mg_probe_placement");
    *ratio = ms_fill > 0.f ? (double)ms_pat / (double)ms_fill : 1.0;
    *pattern_us = 1e3 * (double)ms_pat / reps;
    return MG_OK;
}

extern "C" int mg_device_probe_placement(mg_context *ctx, void *buf, int64_t bytes, double *info) {
    if (!ctx || !buf || !info || bytes < MG_PLACED_MIN_BYTES) {
        mg_set_error("mg_device_probe_placement: needs a context, a buffer of at least 64 MiB and info4");
        return MG_ERR_INVALID_ARGUMENT;
    }
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    double ratio = 1.0, us = 0.0;
    int rc = mg_probe_placement(ctx, buf, bytes, &ratio, &us);
    if (rc != MG_OK) return rc;
    info[0] = 1.0; info[1] = ratio; info[2] = us; info[3] = mg_placement_is_fast(bytes, ratio, us, MG_PLACED_FAST_RATIO, MG_PLACED_CLASS_TBPS) ? 1.0 : 0.0;
    return MG_OK;
}

// What the arena knows about memory it handed out, WITHOUT probing again: info4 = {1 if p is a piece of a placed region,
// pattern / fill of the region's scan, its pattern time in us, 1 if the region is in the fast class}; tbps (may be NULL): the
// pattern's rate in TB/s for the probe that classified the region (0 where the probe was too small to fill the chip).
extern "C" int mg_device_placement_info(mg_context *ctx, const void *buf, double *info, double *tbps) {
    if (!ctx || !buf || !info) { mg_set_error("mg_device_placement_info: needs a context, a pointer and info4"); return MG_ERR_INVALID_ARGUMENT; }
    info[0] = 0.0; info[1] = 1.0; info[2] = 0.0; info[3] = 0.0;
    if (tbps) *tbps = 0.0;
    for (const auto &r : ctx->out_regions)
        if (r.base && (const char *)buf >= r.base && (const char *)buf < r.base + r.bytes) {
            info[0] = 1.0; info[1] = r.ratio; info[2] = r.us; info[3] = r.fast ? 1.0 : 0.0;
            if (tbps) *tbps = r.tbps;
            return MG_OK;
        }
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The context's OUTPUT ARENA.  Every large device allocation the library makes or hands out -- mg_device_malloc from
// MG_PLACED_MIN_BYTES on, mg_device_malloc_placed, the scratch block behind the *_host entry points -- is a piece of a
// PLACED REGION: a buffer that went through the probe once.  Regions stay with the context when their pieces are freed
// (mg_context_trim_outputs releases the idle ones), so an adaptor that allocates its (B, T, D) output on every call pays
// for placement once, and every caller gets the class the bench's buffer gets.
//
// The scan for a new region is bounded so that processes sharing a GPU do not starve each other: at most 32 candidates by
// default, and the candidates held at any moment (rejected ones are held so that the next one comes from other memory)
// never exceed a quarter of the memory that was free when the scan began -- the oldest rejects are released first.
// ---------------------------------------------------------------------------------------------------------------------
#define MG_REGION_GRANULE ((size_t)2 << 20)

static void mg_region_release(mg_context *ctx, mg_context::out_region &r) {
    if (!r.base) return;
    if (r.vmm) (void)mg_device_free_raw(ctx, r.base); else (void)hipFree(r.base);
    r.base = nullptr;
}

// a new region of `bytes` (already a multiple of the granule): probe candidates, keep the first fast one (or the best)
// (probe_bytes: the request itself -- the probe replays the store pattern of a dense (n, 156, 79) block of that size; one tile more
// or less changes the units per workgroup and with them what the pattern measures)
static int mg_region_create(mg_context *ctx, size_t bytes, size_t probe_bytes, int32_t max_candidates, mg_context::out_region *out) {
    size_t free_b = 0, total_b = 0;
    MG_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    const int budget = max_candidates > 0 ? max_candidates : 32;
    size_t hold_cap = std::max<size_t>(free_b / 4, 2 * bytes);   // bytes held at once, the best candidate included
    if (ctx->opt[MG_OPT_PLACED_HOLD] > 0) hold_cap = std::max<size_t>((size_t)ctx->opt[MG_OPT_PLACED_HOLD], 2) * bytes;   // (tests: drops in mid-scan)
    const double fast_ratio = ctx->opt[MG_OPT_PLACED_FAST_PCT] > 0 ? ctx->opt[MG_OPT_PLACED_FAST_PCT] / 100.0 : MG_PLACED_FAST_RATIO;
    struct cand { void *p; bool vmm; };
    std::vector<cand> held;   // rejected candidates, oldest first
    cand best = {nullptr, false};
    double best_ratio = 0.0, best_us = 0.0;
    bool best_fast = false;
    int probed = 0, rc = MG_OK;
    auto drop = [&](const cand &c) { if (c.vmm) (void)mg_device_free_raw(ctx, c.p); else (void)hipFree(c.p); };
    auto consider = [&](void *p, bool vmm) -> bool {   // probe p, keep the better of (best, p), hold the other; false: stop
        double ratio = 1.0, us = 0.0;
        rc = mg_probe_placement(ctx, p, (int64_t)probe_bytes, &ratio, &us);
        if (rc != MG_OK) { drop({p, vmm}); return false; }
        probed++;
        // (every candidate is probed with the same probe_bytes: the pattern's own time orders them; the ratio where the probe is too
        // small to time)
        if (!best.p || (us > 0.0 && best_us > 0.0 ? us < best_us : ratio < best_ratio)) {
            if (best.p) held.push_back(best);
            best = {p, vmm}; best_ratio = ratio; best_us = us;
            best_fast = mg_placement_is_fast((int64_t)probe_bytes, ratio, us, fast_ratio);
        } else {
            held.push_back({p, vmm});
        }
        while (!held.empty() && (held.size() + 2) * bytes > hold_cap) {   // + the best one + the candidate to come
            (void)hipStreamSynchronize(ctx->stream);
            drop(held.front());
            held.erase(held.begin());
        }
        return !best_fast;
    };
    int limit = budget;
    auto plain = [&](int count) {
        for (int i = 0; i < count && probed < limit; i++) {
            void *p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return; }   // out of memory: settle for the best so far
            if (!consider(p, false)) return;
        }
    };
    // Two recipes.  Plain allocations first (on most boxes the second or third is fast); where sixteen of them were not,
    // twelve buffers assembled from physical chunks through the virtual-memory API (on some boxes only those are fast), then
    // the rest of the budget plain again.
    plain(std::min(budget, 16));
    if (best.p && !best_fast && rc == MG_OK && max_candidates != 1) {
        static const int64_t chunk_mib[3] = {32, 8, 2};
        bool go = true;
        for (int c = 0; c < 3 && go; c++)
            for (int i = 0; i < 4 && go && probed < budget; i++) {
                void *p = nullptr;
                if (mg_device_malloc_chunked(ctx, (int64_t)bytes, chunk_mib[c] << 20, &p) != MG_OK) { (void)hipGetLastError(); go = false; break; }
                go = consider(p, true);
            }
        if (rc == MG_OK && !best_fast) plain(budget - probed);
    }
    // The deep scan.  Fast-class memory is SPARSE: of 400 plain allocations of 404 MB held together on one box 14 were fast
    // (tools/probes/deep_scan.py: the first at the 16th, seven among the 26th .. 50th, none among the 101st .. 200th), so a scan of 32
    // misses it on one box in three -- the "slow boxes" of rounds 2-4, 0.54 of the roofline instead of 0.64.  Where the caller set no
    // budget of its own and the first 32 candidates were all slow, the scan goes on, plain allocations only, until a fast one turns
    // up, MG_PLACED_DEEP_CANDIDATES have been probed, or the candidates held (they must stay allocated, or the allocator hands the
    // same memory out again) reach the hold cap: ~12 ms per candidate, once per region.
    // (Round 5: fast-class memory comes in CLUSTERS of the allocation order -- on a box whose first 180 candidates were all slow, 19 of the next 45 were
    // fast, 26 of 600 in all (profiles/r05_frames/deep_scan_600.log) -- and a limit of 160 candidates / a quarter of the memory stopped just short of
    // them on one process in ten (step 94 instead of 77 us).  The deep scan now goes to 400 candidates and may hold six tenths of the free memory while
    // it runs -- for seconds, once per region; everything but the winner is released before it returns.)
    if (rc == MG_OK && best.p && !best_fast && max_candidates == 0 && ctx->opt[MG_OPT_PLACED_HOLD] == 0) {
        limit = MG_PLACED_DEEP_CANDIDATES;
        hold_cap = std::max<size_t>(hold_cap, free_b / 10 * 6);
        while (probed < limit && !best_fast && rc == MG_OK && (held.size() + 3) * bytes <= hold_cap) {
            const int before = probed;
            plain(1);
            if (probed == before) break;        // out of memory
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (const cand &c : held) drop(c);
    if (rc != MG_OK) {   // a probe failed: nothing is kept, the error travels up
        if (best.p) drop(best);
        return rc;
    }
    if (!best.p) {
        mg_set_error("mg_device_malloc: out of device memory (%lld bytes)", (long long)bytes);
        return MG_ERR_OUT_OF_MEMORY;
    }
    out->base = (char *)best.p; out->bytes = bytes; out->vmm = best.vmm; out->ratio = best_ratio; out->us = best_us;
    out->tbps = mg_placement_tbps((int64_t)probe_bytes, best_us);
    // the class: from the scan's own measurement, with the lower threshold (see MG_PLACED_CLASS_TBPS), and it stays
    out->probed = probed; out->fast = best_fast || mg_placement_is_fast((int64_t)probe_bytes, best_ratio, best_us, fast_ratio, MG_PLACED_CLASS_TBPS);
    out->live = 0;
    out->free_list.clear();
    out->free_list.push_back({0, bytes});
    return MG_OK;
}

// a piece of `bytes` from a placed region: first fit in the regions the context has, else a new region of that size
int mg_output_alloc(mg_context *ctx, int64_t bytes, int32_t max_candidates, void **out, double *info) {
    const size_t need = ((size_t)bytes + MG_REGION_GRANULE - 1) / MG_REGION_GRANULE * MG_REGION_GRANULE;
    for (int pass = 0; pass < 2; pass++) {
        for (auto &r : ctx->out_regions) {
            if (pass == 0 && !r.fast) continue;   // fast regions first
            for (size_t i = 0; i < r.free_list.size(); i++) {
                if (r.free_list[i].second < need) continue;
                *out = r.base + r.free_list[i].first;
                r.used[(char *)*out - r.base] = need;
                if (r.free_list[i].second == need) r.free_list.erase(r.free_list.begin() + (long)i);
                else { r.free_list[i].first += need; r.free_list[i].second -= need; }
                r.live++;
                if (info) { info[0] = 0.0; info[1] = r.ratio; info[2] = r.us; info[3] = r.fast ? 1.0 : 0.0; }
                return MG_OK;
            }
        }
    }
    // No region can serve the request.  Idle regions that are too small for it would stay reserved beside the new one for as long
    // as the workload's sizes keep growing: they go first.
    bool released = false;
    for (size_t i = ctx->out_regions.size(); i-- > 0;)
        if (ctx->out_regions[i].live == 0) {
            if (!released) { (void)hipStreamSynchronize(ctx->stream); released = true; }
            mg_region_release(ctx, ctx->out_regions[i]);
            ctx->out_regions.erase(ctx->out_regions.begin() + (long)i);
        }
    mg_context::out_region r;
    int rc = mg_region_create(ctx, need, (size_t)bytes, max_candidates, &r);
    if (rc == MG_ERR_OUT_OF_MEMORY) {
        // not even one candidate of the scan could be allocated: the last resort is what mg_device_malloc was before there were
        // regions -- ONE allocation, unprobed (class unknown: reported as slow)
        void *p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) {
            (void)hipGetLastError();
            mg_set_error("mg_device_malloc: out of device memory (%lld bytes)", (long long)need);
            return MG_ERR_OUT_OF_MEMORY;
        }
        r = mg_context::out_region();
        r.base = (char *)p; r.bytes = need; r.vmm = false; r.fast = false; r.ratio = 1.0; r.us = 0.0; r.probed = 0;
        rc = MG_OK;
    }
    if (rc != MG_OK) return rc;
    *out = r.base;
    r.used[0] = need;
    r.free_list.clear();
    r.live = 1;
    if (info) { info[0] = r.probed; info[1] = r.ratio; info[2] = r.us; info[3] = r.fast ? 1.0 : 0.0; }
    ctx->out_regions.push_back(std::move(r));
    return MG_OK;
}

// true if p was a piece of a region (and is now free again)
bool mg_output_free(mg_context *ctx, void *p) {
    for (auto &r : ctx->out_regions) {
        if ((char *)p < r.base || (char *)p >= r.base + r.bytes) continue;
        const size_t off = (size_t)((char *)p - r.base);
        auto it = r.used.find(off);
        if (it == r.used.end()) return false;
        std::pair<size_t, size_t> blk = {off, it->second};
        r.used.erase(it);
        r.live--;
        // insert sorted, merge with neighbours
        size_t i = 0;
        while (i < r.free_list.size() && r.free_list[i].first < blk.first) i++;
        r.free_list.insert(r.free_list.begin() + (long)i, blk);
        if (i + 1 < r.free_list.size() && r.free_list[i].first + r.free_list[i].second == r.free_list[i + 1].first) {
            r.free_list[i].second += r.free_list[i + 1].second;
            r.free_list.erase(r.free_list.begin() + (long)i + 1);
        }
        if (i > 0 && r.free_list[i - 1].first + r.free_list[i - 1].second == r.free_list[i].first) {
            r.free_list[i - 1].second += r.free_list[i].second;
            r.free_list.erase(r.free_list.begin() + (long)i);
        }
        return true;
    }
    return false;
}

void mg_output_release_all(mg_context *ctx) {
    for (auto &r : ctx->out_regions) mg_region_release(ctx, r);
    ctx->out_regions.clear();
}

extern "C" int mg_context_trim_outputs(mg_context *ctx) {
    if (!ctx) { mg_set_error("mg_context_trim_outputs: ctx is NULL"); return MG_ERR_INVALID_ARGUMENT; }
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (size_t i = ctx->out_regions.size(); i-- > 0;)
        if (ctx->out_regions[i].live == 0) {
            mg_region_release(ctx, ctx->out_regions[i]);
            ctx->out_regions.erase(ctx->out_regions.begin() + (long)i);
        }
    return MG_OK;
}

// the placement class of memory the arena handed out: 1 fast, 0 slow, -1 unknown (not a piece of a placed region)
int mg_output_class(mg_context *ctx, const void *p) {
    if (!ctx || !p) return -1;
    for (const auto &r : ctx->out_regions)
        if (r.base && (const char *)p >= r.base && (const char *)p < r.base + r.bytes) return r.fast ? 1 : 0;
    return -1;
}

extern "C" int mg_context_output_bytes(mg_context *ctx, int64_t *reserved, int64_t *in_use, int32_t *n_regions, int32_t *n_fast) {
    if (!ctx) { mg_set_error("mg_context_output_bytes: ctx is NULL"); return MG_ERR_INVALID_ARGUMENT; }
    int64_t res = 0, use = 0;
    int32_t nf = 0;
    for (auto &r : ctx->out_regions) {
        res += (int64_t)r.bytes;
        for (auto &u : r.used) use += (int64_t)u.second;
        nf += r.fast ? 1 : 0;
    }
    if (reserved) *reserved = res;
    if (in_use) *in_use = use;
    if (n_regions) *n_regions = (int32_t)ctx->out_regions.size();
    if (n_fast) *n_fast = nf;
    return MG_OK;
}

extern "C" int mg_device_malloc_placed(mg_context *ctx, int64_t bytes, int32_t max_candidates, void **out_dev, double *info) {
    if (!ctx || !out_dev || bytes < 0) {
        mg_set_error("mg_device_malloc_placed: bad arguments");
        return MG_ERR_INVALID_ARGUMENT;
    }
    *out_dev = nullptr;
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    if (info) { info[0] = 0.0; info[1] = 1.0; info[2] = 0.0; info[3] = 1.0; }
    if (bytes < MG_PLACED_MIN_BYTES) return mg_device_malloc(ctx, bytes, out_dev);
    return mg_output_alloc(ctx, bytes, max_candidates, out_dev, info);
}

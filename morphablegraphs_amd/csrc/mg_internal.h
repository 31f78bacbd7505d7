// Internal declarations shared by the libmg_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/mg_hip.h"

#define MG_NCAND 16         // candidates per MFMA tile (N of v_mfma_f32_16x16x4_f32)
#define MG_ROWTILE 16       // coefficient rows per MFMA tile (M)
#define MG_MAX_KK 16        // k-steps of 4 -> n_components <= 64 on the MFMA path
#define MG_BLOCK 256        // threads per workgroup of the frames kernels
#define MG_PROFILE_SLOTS 11

void mg_set_error(const char *fmt, ...);
int mg_hip_fail(hipError_t e, const char *what);

#define MG_HIP_CHECK(expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return mg_hip_fail(_e, #expr); \
    } while (0)

struct mg_event_pair {
    hipEvent_t a, b;
    int slot;
    bool ended;
    int slot2 = -1;   // the same duration counted into a second slot as well (the fused step: "frames" and "step")
};

struct mg_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 0;
    int64_t total_mem = 0;
    int max_lds = 0;
    char name[256] = {0};
    // profiling
    bool profile = false;
    std::vector<mg_event_pair> pending;
    std::vector<hipEvent_t> free_events;
    int reserved_cus = 0;                        // CUs the persistent kernel leaves free (e.g. for RCCL kernels)
    int32_t opt[MG_OPT_COUNT] = {0};             // mg_context_set_option (test / tuning switches, 0 = default)
    int prof_interval = 1;                       // bracket every n-th launch of a slot
    int64_t prof_seen[MG_PROFILE_SLOTS] = {0};
    double prof_ms[MG_PROFILE_SLOTS] = {0};
    int64_t prof_n[MG_PROFILE_SLOTS] = {0};
    std::vector<float> prof_samples[MG_PROFILE_SLOTS];   // individual durations, first 65536 per slot
    // scratch for host convenience variants / argmin
    void *scratch = nullptr;
    int64_t scratch_bytes = 0;
    void *argmin_out = nullptr;  // 16 bytes device
    // device arena for long-lived constants (mg_context_arena_begin/_end): blocks are bump-allocated, freed with the context
    struct arena_block { char *base; size_t bytes, used; int64_t live; };   // live: arrays handed out and not yet freed
    std::vector<arena_block> arena;
    size_t arena_block_bytes = 0;   // > 0 while an arena section is open
    struct vmm_alloc { void *va; size_t total, chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
    std::vector<vmm_alloc> vmm;     // buffers from mg_device_malloc_chunked
    // Address ranges of released chunked buffers.  Their physical memory is gone, the reservation stays until the context is
    // destroyed: a range the runtime hands out again for ANOTHER mapping while translations of the old one are still cached
    // makes kernels store through the stale ones (profiles/r03_placement/t8_remap_stale_translation.log: "STORES LOST").
    std::vector<std::pair<void *, size_t>> vmm_parked;
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};   // mg_options_step: the options of a step are independent chains of small
    hipEvent_t side_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // launches; four of them run side by side ([4]: the fork)
    // mg_options_step as one launch (mg_options.hip): the per-option constants on the device and their host copy, the
    // per-option arrival counters and one {value, index} partial per workgroup
    void *fused_tab_dev = nullptr;
    std::vector<unsigned char> fused_tab_host;
    void *fused_counters = nullptr, *fused_partials = nullptr;
    void *fused_dyn_dev = nullptr;   // a step's per-option values where the device draws the component counts itself
    // ... two slots of them: the step's kernel also draws the counts the NEXT step will want if its seeds are this step's + 1 (a planner
    // counts its steps), so that step needs no counts kernel in front; what was drawn for whom:
    struct { bool valid = false; int32_t n_options = 0, slot = 0; int64_t n = 0; uint64_t seeds[24] = {0}; const void *prims[24] = {nullptr}; } fused_next;
    unsigned attr_gmm_lds = 0, attr_traj = 0;   // dynamic-LDS attributes already set for this context's device (bit per instantiation)
    unsigned long long fused_seq = 0;   // sequence number of the planner steps whose records the kernel leaves in pinned memory
    int fused_partials_n = 0;
    // the output arena (mg_placement.hip): buffers that went through the placement probe, sub-allocated in 2 MiB granules
    struct out_region {
        char *base = nullptr; size_t bytes = 0; bool vmm = false, fast = false; double ratio = 1.0, us = 0.0, tbps = 0.0; int probed = 0;
        std::vector<std::pair<size_t, size_t>> free_list;   // (offset, bytes), sorted by offset
        std::map<size_t, size_t> used;                       // offset -> bytes of the pieces handed out
        int64_t live = 0;
    };
    std::vector<out_region> out_regions;
    void *pinned = nullptr;         // pinned host staging block for small read-backs
    size_t pinned_bytes = 0;
    // Completion flags of the planner step's kernel (mg_wait_flags): a pinned block of their OWN that nothing else is ever written
    // to -- inside the staging block they sat behind the records, at an offset that moves with the option count, where an earlier
    // step's record bytes could read as this step's sequence number.
    unsigned long long *flag_block = nullptr;
    size_t flag_cap = 0;            // flags
    void *traj_paths = nullptr;     // mg_score_trajectory[ies]: the candidates' root paths (B, T, 3) float64 for the reference's search
    size_t traj_paths_bytes = 0;    // (kept between calls, grown on demand, freed with the context)
    void *lists_dev = nullptr;      // mg_options_frame_lists: per-option argument tables, workgroup partials, arrival counters
    size_t lists_bytes = 0, lists_counters_off = 0;
    void *rccl_comm = nullptr;      // ncclComm_t after mg_dist_init
    int dist_rank = 0, dist_ranks = 1;
};
void mg_dev_free(mg_context *ctx, void *p);   // hipFree unless p lives in the context's arena

int mg_ctx_scratch(mg_context *ctx, int64_t bytes, void **out);
void mg_prof_begin(mg_context *ctx, int slot);
// Events attached to ONE kernel launch (hipExtLaunchKernel start / stop events: the dispatch's own timestamps, no
// marker packets between launches): true and two events if this launch of `slot` is to be timed, else false.
bool mg_prof_kernel(mg_context *ctx, int slot, int slot2, hipEvent_t *start, hipEvent_t *stop);
void mg_prof_end(mg_context *ctx, int slot);
int mg_prof_resolve(mg_context *ctx);

// One chunk of a time grid handled by one workgroup of the MFMA kernel.
#define MG_MAX_WI 11         // basis functions per chunk window (11 x 3 root rows still span <= 3 row tiles of 16)
#define MG_MAX_NT 48         // time samples per chunk
#define MG_TAP_KS ((MG_MAX_WI + 3) / 4)   // k-steps of the banded root-tap MFMA (4 basis functions each)
#define MG_TAP_FT (MG_MAX_NT / 16)       // sample tiles of 16 of the root-tap MFMA
#define MG_ARG_CHUNKS 8   // chunk descriptors that travel in the frames kernels' arguments
// LDS of the frames kernels per ring slot: per-sample tables (tap weights + tap row offsets) and the root outputs
// [candidate][sample][4] float32.  nt = the grid's longest chunk, rounded up to 16 samples.  A candidate's root outputs are
// MG_RO_PAD floats apart from a multiple of 32: wave 0 writes them one float per lane, (candidate, channel) x sample, and with a
// stride of 4 nt floats (0 mod 32 banks for nt = 16, 32, 48) the six candidates of a column tile meet in the same banks.
#ifndef MG_RO_PAD
#define MG_RO_PAD 8
#endif
#define MG_RO_CS(nt) ((nt) * 4 + MG_RO_PAD)              // floats per candidate
#define MG_TB_BYTES_N(nt) ((nt) * 16 + (nt) * 4)
#define MG_RO_BYTES_N(nt) (MG_NCAND * MG_RO_CS(nt) * 4)
struct mg_chunk {
    int32_t t0;        // first time index (chunks are runs of consecutive time indices)
    int32_t nT;        // number of time samples in the chunk, <= MG_MAX_NT
    int32_t rt0;       // first global 16-row tile of the (padded-row) coefficient window
    int32_t ntiles;    // number of 16-row tiles
    int32_t imin;      // first coefficient index of the window
    int32_t wi;        // coefficient rows (basis functions) in the window, <= MG_MAX_WI
    int32_t rrt0;      // first 16-row tile of the root-row window (rows = i*nroot + d)
    int32_t nrt;       // number of root tiles
};

struct mg_time_grid {
    mg_primitive *prim = nullptr;
    bool owned_by_prim = false;
    int32_t T = 0;
    std::vector<double> times;
    std::vector<int32_t> i0;
    std::vector<double> w;  // (T,4)
    std::vector<mg_chunk> chunks;
    // device tables
    int32_t *d_i0 = nullptr;
    double *d_w = nullptr;      // (T,4) float64 weights
    float *d_w32 = nullptr;     // (T,4) float32 weights
    float *d_rootm = nullptr;   // (T,8) the root channels' mean part M[f][d] = spline of mean' alone (float64 on the host) as
                                // float32 pairs: {Mhi[0..2], Mlo[0..2], 0, 0} (the mean/delta split of the frames kernels)
    double *d_wtap = nullptr;   // [n_chunks][2][2][64] banded tap weights, f64 MFMA A fragments
    mg_chunk *d_chunks = nullptr;
    int32_t n_chunks = 0;
    int32_t stride = 0;      // floats per candidate in the LDS coefficient image
    int32_t max_wi = 0;
    int32_t max_nt = 16;        // samples of the longest chunk, rounded up to 16: sizes the per-slot tables in LDS
    int32_t lds_bytes = 0;   // dynamic LDS of the MFMA kernel for this grid
    int32_t nbuf = 2;        // LDS ring depth of the MFMA kernel (3 when it fits)
    bool mfma_ok = false;
    int32_t max_tiles = 0;   // row tiles of the widest chunk window
    int32_t cs_lds_bytes = 0;   // dynamic LDS of the chunk-stationary kernel (without the fused mixture's buffers)
    bool cs_ok = false;      // the chunk-stationary kernel covers this grid (window fits the row producers' registers, LDS fits)
};

struct mg_primitive {
    mg_context *ctx = nullptr;
    int32_t NB = 0, D = 0, L = 0, F = 0, K = 0, R = 0;
    int32_t nroot = 0;  // min(3, D): root-translation channels (float64 pipeline or mean/delta split)
    bool root_split = false;      // the accuracy gate allows the mean/delta split (mg_primitive_root_mode)
    double root_split_est = 0.0;  // its error estimate
    int32_t KK = 0;     // MFMA k-steps (even), 0 when L > 64
    int32_t Lg = 0;     // dimension of the mixture (>= L: spatial + time latents)
    int32_t KKg = 0;    // MFMA k-steps of the mixture kernels (even), 0 when Lg > 64
    int32_t Lt = 0, NBt = 0;         // time model: components, basis functions
    double *d_tphi = nullptr;        // [F][Lt]: harmonics at the canonical frames
    double *d_tmean = nullptr;       // [F]: mean time spline at the canonical frames
    int32_t RT = 0;     // 16-row tiles of the padded-row space NB*Dp
    // host float64 copies (already scaled by translation_maxima)
    std::vector<double> Es;    // (R, L)
    std::vector<double> means_;  // mean' (R)
    std::vector<double> knots;
    std::vector<double> gw, gm, gc, gp;  // weights (K), means (K,L), covars (K,L,L), prec chol (K,L,L)
    // device constants
    int32_t Dp = 0;              // row pitch of the padded coefficient rows (multiple of 4)
    int32_t cshift = 0;          // column of channel d in a padded row is d + cshift
    float *d_Epack = nullptr;    // [RT][KK/2][64][2] MFMA A fragments, f32, padded rows r' = i*Dp + d
    float *d_Et32 = nullptr;     // [L][R] f32
    double *d_Et64 = nullptr;    // [L][R] f64
    double *d_Erpack = nullptr;  // [RRT][KK][64] f64 MFMA A fragments of the root rows (row = i*nroot + d)
    double *d_meanroot = nullptr;  // [RRT*16] f64 mean' of the root rows
    float *d_mean32 = nullptr;   // [RT*16] f32 mean' in padded-row order (zero padded; zero on the root rows as well, which
                                 // therefore hold E'.s alone: the delta of the mean/delta split)
    double *d_mean = nullptr;    // (R) f64
    double *d_knots = nullptr;   // (NB + 4) f64: the spatial knot vector (basis rows computed on the device: mg_frames_at_kernel)
    int32_t RRT = 0;             // 16-row tiles of the root-row space
    // GMM device constants
    double *d_gPpack = nullptr;  // [K][JT][KK][64]: v_mfma_f64_16x16x4_f64 B fragments of P_k (L <= 64)
    double *d_gmPpad = nullptr;  // [K][JT*16]: mu_k . P_k, zero padded
    double *d_gP = nullptr;      // [K][L(j)][L(i)]: column j of P_k contiguous over i
    double *d_gmP = nullptr;     // [K][L]: mu_k . P_k
    double *d_gPTpack = nullptr; // [K][JT][KK][64]: B fragments of P_k^T (Jacobian)
    double *d_gcholpack = nullptr; // [K][JT][KK][64]: B fragments of chol_k^T (sampler)
    double *d_gmeanpad = nullptr;  // [K][JT*16]: means, zero padded
    double *d_gconst = nullptr;  // [K]: log w_k + sum log diag P_k - 0.5 L log 2pi
    double *d_gmean = nullptr;   // [K][L]
    double *d_gchol = nullptr;   // [K][L][L] lower Cholesky of covars (sampler)
    mg_time_grid *canonical = nullptr;
    mg_time_grid *coeff_grid = nullptr;  // identity basis: "frames" == coefficients
};

struct mg_constraint_set {
    mg_primitive *prim = nullptr;
    int32_t n = 0;
    int32_t nch = 0;            // pose channels of the primitive the root constraints may touch (<= 7)
    double *d_W = nullptr;      // [rows][L]: fused keyframe matrices, rows of constraint c start at woff[c]
    double *d_bias = nullptr;   // [rows]
    double *d_par = nullptr;    // [n][8]: type, weight, target[3], ref_dir[3]
    int32_t *d_woff = nullptr;  // [n + 1] first row of every constraint
    int32_t *d_chain = nullptr; // [n] FK chain length m (0 for root constraints); midpoint: m | m2 << 16
    double *d_choff = nullptr;  // [n][2][MG_MAX_CHAIN][3] offsets along the chain (second chain: midpoint only)
    double *d_Wpack = nullptr;  // [RT][KK][64] MFMA B fragments of W (n_components <= 64), RT = ceil(rows / 16)
    double *d_bpad = nullptr;   // [RT*16] bias, zero padded
    int32_t RT = 0;
    int32_t rows = 0;           // rows of W in use (RT = ceil(rows / 16))
    std::vector<mg_keyframe_constraint> structure;   // what mg_constraint_set_update must find unchanged
    int32_t align_joint = -1;   // -1: local mode
    double *d_pose = nullptr;   // pose constraints' tables (MG_POSE_* layout), NULL if there are none
    bool has_pose = false;
    double *d_align = nullptr;  // [8] chain length, previous heading (x, z), previous root (x, z), ref_dir; NULL = local mode;
                                // its rows (first control point: root xyz, then the chain's quaternions) start at woff[n]
};

// Device table of one pose constraint inside d_pose (doubles): header, then one record per point
#define MG_POSE_HDR 8                            // [0] n_points, [1] has_velocity, [2..4] velocity, [5] rows of one pose block
#define MG_POSE_REC (5 + 4 * MG_MAX_CHAIN)       // target xyz, weight, chain length m, then m x (quaternion row or -1, offset xyz)

// launchers (each validates nothing: the C-ABI entry points did)
int mg_launch_frames_mfma(mg_primitive *p, const mg_time_grid *g, const void *lat, int ldt, int64_t B, int64_t ld, float *out, float *logp,
                          int prof_slot = -1, int prof_slot2 = -1);   // prof_slot >= 0: the launch carries its own timing events
bool mg_frames_can_fuse_gmm(const mg_primitive *p, const mg_time_grid *g, int64_t B);
int mg_frames_kernel_choice(const mg_primitive *p, const mg_time_grid *g, int64_t B, bool fused, const void *out = nullptr, bool lat_f64 = false);
int mg_output_class(mg_context *ctx, const void *p);   // mg_placement.hip: 1 fast, 0 slow (a piece of a placed region), -1 not the arena's
int mg_frames_lds_bytes(const mg_primitive *p, const mg_time_grid *g, int which, bool fused);
int mg_frames_grid(const mg_primitive *p, const mg_time_grid *g, int64_t B, int which);
bool mg_frames_root_split(const mg_primitive *p);   // the root-channel mode a launch uses (gate + MG_OPT_ROOT_MODE)
#define MG_ROOT_SPLIT_MAX_EST 5e-6
int mg_cs_max_tiles(int KK);   // row tiles of a chunk window the chunk-stationary kernel can hold in registers
int mg_launch_frames_direct(mg_primitive *p, const mg_time_grid *g, const void *lat, int ldt, int64_t B, int64_t ld, void *out, bool out_f64);
int mg_launch_spline_eval(mg_primitive *p, const mg_time_grid *g, const double *coeffs, int64_t n, double *out);
int mg_launch_gmm_logp(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, void *out, int odt);
bool mg_objective_can_fuse(const mg_primitive *p, const mg_constraint_set *cs);
int mg_launch_objective(mg_primitive *p, const mg_constraint_set *cs, const void *x, int xdt, int64_t B, int64_t ld, double error_scale,
                        double quality_scale, double *logp_out, double *err_out, double *obj_out);
#define MG_SAMPLE_ARG_K 16   // mixtures up to this size pass their prefix sums to the sampler as a kernel argument
int mg_launch_gmm_sample(mg_primitive *p, int64_t n, const int64_t *cum_dev, const int64_t *cum_host, int64_t n_tiles, uint64_t seed, void *x, int xdt, int64_t ld, int32_t *comp,
                         int64_t tile0, int64_t row_lo, int64_t row_hi);
bool mg_gmm_sample_takes_host_prefix(const mg_primitive *p);
int mg_launch_set_params(mg_context *ctx, const double *values, int n_par, int n_align, double *d_par, double *d_align);
int mg_launch_score(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int ldt, int64_t B, int64_t ld, void *out, int odt, double *res,
                    const double *align_cand = nullptr);
int mg_launch_gather_winner(mg_context *ctx, const void *x, int xdt, int64_t ld, int L, void *result_dev);
int mg_launch_argmin_gather(mg_context *ctx, const void *v, int dt, int64_t n, void *result_dev, const void *x, int xdt, int64_t ld, int L, int64_t index_offset = 0);
int mg_launch_gmm_jac(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, double *out);
int mg_launch_time_function(mg_primitive *p, const void *gamma, int gdt, int64_t B, int64_t ld, double *out);
int mg_launch_timewarp(mg_primitive *p, const void *gamma, int gdt, int64_t B, int64_t ld, double speed, double *times, int32_t *lens, int32_t t_cap,
                       double *canonical_out);   // mg_timewarp.hip
int mg_launch_frames_at(mg_primitive *p, const void *lat, int ldt, int64_t B, int64_t ld, const double *times, const int32_t *lens, int32_t t_cap,
                        void *out, int odt);
int mg_launch_argmin(mg_context *ctx, const void *v, int dt, int64_t n, void *out_dev);
int mg_launch_joint_positions(mg_context *ctx, const double *frames, const double *table, int64_t N, int D, int J, double *out);
int mg_setup_kernel_attributes(mg_context *ctx);
int mg_options_fused_attributes();   // mg_options.hip
bool mg_options_can_fuse(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n);
int mg_launch_options_fused(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n,
                            const int64_t *const *counts, const uint64_t *seeds, void *const *x_dev, int xdt, const int64_t *ld,
                            double *const *errors_dev, void *results_dev, int64_t result_stride, int64_t row_begin, int64_t row_count,
                            void *records_host = nullptr, int32_t *counts_host = nullptr, unsigned long long *flags_host = nullptr,
                            unsigned long long seq = 0);
int mg_probe_placement(mg_context *ctx, void *buf, int64_t bytes, double *ratio, double *pattern_us);   // mg_placement.hip
#define MG_PLACED_MIN_BYTES ((int64_t)64 << 20)   // below this an output sits in the 256 MiB Infinity Cache anyway
int mg_output_alloc(mg_context *ctx, int64_t bytes, int32_t max_candidates, void **out, double *info4);   // a piece of a placed region
bool mg_output_free(mg_context *ctx, void *p);
void mg_output_release_all(mg_context *ctx);
int mg_device_free_raw(mg_context *ctx, void *p);   // hipFree / virtual-memory release, no arena lookup

// host-side float64 spline basis (FITPACK splev/fpbspl semantics)
void mg_basis_row(const double *knots, int n_knots, double x, int32_t *i0, double *w4);

// A trajectory constraint's target spline on the device (mg_trajectory.hip; read by mg_frame_constraints.hip as well)
struct mg_trajectory {
    mg_primitive *prim = nullptr;
    int32_t n_seg = 0, granularity = 1000, rows = 0;
    double *d_poly = nullptr;    // [n_seg][4][3]: point(t) = ((A0 t + A1) t + A2) t + A3 on a segment; then the last control point [3]
    double *d_arc = nullptr;     // [granularity + 1]: relative arc length at u = k / granularity (arc_length_map.py:45-71)
    double full_arc = 0.0;
    double *d_E = nullptr;       // [rows][L]: root coefficient rows (i * 3 + d), then the first control point's root quaternion (4 rows)
    double *d_mean = nullptr;    // [rows]
};

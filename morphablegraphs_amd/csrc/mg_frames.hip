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
//   channels d <  nroot : either the same in float64 (v_mfma_f64_16x16x4_f64 / fma), out = (float)v64,
//       or -- where the primitive's accuracy gate allows it (mg_primitive_root_mode) -- the mean/delta split:
//       delta = the float32 pipeline above with C-in = 0 (the root rows of the ordinary row tiles),
//       out = Mhi[f][d] + (Mlo[f][d] + delta), (Mhi, Mlo) = the float64 spline of mean' alone as a float32 pair
//       (a table of the time grid): no float64 arithmetic on the device.
//
// This translation unit: which kernel a launch uses (mg_frames_kernel_choice), its grid and LDS, the dispatch.  The kernels:
// mg_frames_cs.hip (chunk-stationary), mg_frames_ws.hip (tile-major), mg_frames_direct.hip (one thread per element).
// The diagnostic build (-DMG_DEBUG_BUILD: ablation switches, phase timers in device globals) compiles them all as ONE
// translation unit -- this file includes the others -- so that they share the timers' device arrays.
#include "mg_frames_common.h"
#ifdef MG_DEBUG_BUILD
#include "mg_frames_ws.hip"
#include "mg_frames_cs.hip"
#include "mg_frames_direct.hip"
#endif

int mg_setup_kernel_attributes(mg_context *) {
    int rc = mg_frames_ws_attributes();
    return rc != MG_OK ? rc : mg_frames_cs_attributes();
}

bool mg_frames_root_split(const mg_primitive *p) {
    const int want = p->ctx->opt[MG_OPT_ROOT_MODE];
    // the float64 pipeline unless asked: the split measured SLOWER on the kernels it was meant to speed up (the sweep
    // waves are issue-bound, its two extra additions and table reads cost more than wave 0's float64 stage, which is off the critical path)
    return want == 2 || (want == 3 && p->root_split);
}

// LDS of the fused mixture scoring: two term buffers and two exp buffers of [K][16] float64
static int mg_fused_gmm_lds(const mg_primitive *p) { return 4 * p->K * 16 * 8; }
// ... and, in the chunk-stationary kernel, what its start-up stages for the tail: the workgroup's (first) two latent tiles as float32 A fragments [2][KK][64]
static int mg_fused_gmm_lds_cs(const mg_primitive *p) {   // + the components' C-in rows [K][JT*16] and constants [K], float64
    const int JT = (int)((p->Lg + 15) / 16);
    return mg_fused_gmm_lds(p) + 2 * p->KK * 64 * 4 + (p->K * JT * 16 + (p->K + 1) / 2 * 2) * 8;
}

// the staged form of the chunk-stationary kernel's mixture tail where its extra LDS fits (otherwise the tail loads everything itself, as before)
static bool mg_fused_gmm_staged(const mg_primitive *p, const mg_time_grid *g) { return g->cs_lds_bytes + mg_fused_gmm_lds_cs(p) <= 160 * 1024; }

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
// out (may be NULL = unknown): where the frames go.  On slow-class memory -- a piece of one of the context's placed regions whose
// scan found nothing better; a whole box can be like that -- the tile-major kernel is the faster one for the large outputs the class
// exists for ('walk', us per step, chunk-stationary / tile-major: B = 8192 95.8 / 93.1 on a slow buffer and 78.4 / 82.8 on a fast
// one in the same process; B = 16 384 195.2 / 186.4 slow), so there the batch-size rule gives way.
int mg_frames_kernel_choice(const mg_primitive *p, const mg_time_grid *g, int64_t B, bool fused, const void *out, bool lat_f64) {
    const int64_t n_tiles = (B + MG_NCAND - 1) / MG_NCAND;
    const int64_t units = n_tiles * g->n_chunks;
    const int64_t grid0 = std::min<int64_t>(units, std::max(1, p->ctx->n_cu - p->ctx->reserved_cus));
    const int64_t grid_cs = grid0 / g->n_chunks * g->n_chunks;   // whole workgroups per chunk
    const int want = p->ctx->opt[MG_OPT_FRAMES_KERNEL];
    bool cs = g->cs_ok && grid_cs >= g->n_chunks && grid_cs <= 4096 &&
              (!fused || (g->cs_lds_bytes + mg_fused_gmm_lds(p) <= 160 * 1024 && (n_tiles + grid_cs - 1) / grid_cs <= 4));
    // (the instantiations whose wave 0 does not fit the register budget -- 61 .. 64 latents; float64 latents from 53 on -- spill inside
    // the unit loop, where scratch traffic queues behind the store stream: those shapes stay with the tile-major kernel unless asked for)
    const bool cs_spills = p->KK >= 16 || (lat_f64 && p->KK >= 14);
    if (want == 2 && !cs) return -1;
    if (want == 1) cs = false;
    else if (want == 0 && cs_spills) cs = false;
    else if (want == 0) cs = cs && units >= 2 * grid0 && mg_output_class(p->ctx, out) != 0;
    return cs ? 2 : 1;
}
int mg_frames_grid(const mg_primitive *p, const mg_time_grid *g, int64_t B, int which) {
    const int64_t units = (B + MG_NCAND - 1) / MG_NCAND * g->n_chunks;
    const int64_t grid0 = std::min<int64_t>(units, std::max(1, p->ctx->n_cu - p->ctx->reserved_cus));
    return (int)(which == 2 ? grid0 / g->n_chunks * g->n_chunks : grid0);
}

int mg_frames_lds_bytes(const mg_primitive *p, const mg_time_grid *g, int which, bool fused) {
    if (which == 2) return g->cs_lds_bytes + (fused ? (mg_fused_gmm_staged(p, g) ? mg_fused_gmm_lds_cs(p) : mg_fused_gmm_lds(p)) : 0);
    return g->lds_bytes + (fused ? mg_fused_gmm_lds(p) : 0);
}

int mg_launch_frames_mfma(mg_primitive *p, const mg_time_grid *g, const void *lat, int ldt, int64_t B, int64_t ld, float *out, float *logp,
                          int prof_slot, int prof_slot2) {
    mg_frames_args a;
    a.rt_total = p->RT;
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
    const int which = mg_frames_kernel_choice(p, g, B, logp != nullptr, out, lf);
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
    a.gmm_staged = cs && logp && mg_fused_gmm_staged(p, g) ? 1 : 0;
    if (logp) lds += a.gmm_staged ? mg_fused_gmm_lds_cs(p) : mg_fused_gmm_lds(p);
    const int grid = mg_frames_grid(p, g, B, which);
    if (cs) {
        const int Q = grid / g->n_chunks;
        a.cs_magic = (1 << 20) / g->n_chunks + 1;
        a.cs_per = (int32_t)(n_tiles / Q);
        a.cs_rem = (int32_t)(n_tiles % Q);
    }
    mg_launch_events ev;
    if (prof_slot >= 0) (void)mg_prof_kernel(p->ctx, prof_slot, prof_slot2, &ev.start, &ev.stop);
    const bool split = mg_frames_root_split(p);
    return cs ? mg_launch_frames_cs(p, g, lat, out, logp, a, lf, split, buf_bytes, lds, grid, ev)
              : mg_launch_frames_ws(p, g, lat, out, logp, a, lf, split, buf_bytes, lds, grid, ev);
}

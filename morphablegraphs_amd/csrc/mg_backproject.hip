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
#include <cstdlib>

#include "mg_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// scalars of one launch; the pointers are separate __restrict__ kernel parameters
struct mg_frames_args {
    int64_t B, ld;
    int32_t T, D, Dp, L, nroot, n_chunks, n_tiles, stride, max_wi;
    int32_t debug;   // MG_DEBUG_FLAGS (bench ablations only): 1 = skip stage 1, 2 = skip stage 2 stores
};

template <bool F64>
__device__ __forceinline__ double mg_load_lat(const void *lat, int64_t idx) {
    if (F64) return ((const double *)lat)[idx];
    return (double)((const float *)lat)[idx];
}

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte store

// One workgroup = 16 candidates x one time chunk (consecutive time samples whose taps fall in
// a window of <= 8 basis functions).
//   stage 1: the window's coefficients for the 16 candidates by v_mfma_f32_16x16x4_f32
//            (A = E' fragments of the padded rows r' = i*Dp + d streamed from L2, next tiles
//            prefetched; B = the latent tile held in registers; C-in = mean'); accumulators ->
//            LDS image [cand][i_local*Dp + d].  Root-translation rows by
//            v_mfma_f64_16x16x4_f64 -> LDS (float64), then their spline taps in float64 ->
//            float32 root outputs in LDS.  The chunk's weights / first taps are staged in LDS, so
//            no memory read follows the first store of the workgroup.
//   stage 2: "quad-row" sweep.  A wave owns one candidate at a time; a lane owns 4 consecutive
//            channels of one time sample (ds_read_b128 per tap, 16 FMAs) and 64/(Dp/4) samples are
//            in flight per wave, so one wave store instruction writes ~1 KB of consecutive
//            bytes and consecutive instructions continue where the last one ended -- the store
//            stream every candidate's (F, D) block wants.
template <int KK, bool LAT_F64>
__global__ __launch_bounds__(MG_BLOCK) void mg_frames_mfma_kernel(
    const float *__restrict__ Epack,      // [RT][KK/2][64][2]
    const float *__restrict__ mean32,     // [RT*16]
    const double *__restrict__ Erpack,    // [RRT][KK][64]
    const double *__restrict__ meanroot,  // [RRT*16]
    const void *__restrict__ lat,         // (B, ld) f32 or f64
    const int32_t *__restrict__ i0tab,    // (T)
    const float4 *__restrict__ w32,       // (T)
    const double *__restrict__ w64,       // (T, 4)
    const mg_chunk *__restrict__ chunks,
    float *__restrict__ out,              // (B,T,D)
    const mg_frames_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware block -> (tile, chunk): blocks that share blockIdx % 8 (one XCD under
    // round-robin dispatch; a speed assumption only) take all chunks of the same tiles, so a
    // candidate's neighbouring output ranges are written through one L2.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int tile = (slot / a.n_chunks) * 8 + xcd;
    const int chunk_id = slot % a.n_chunks;
    if (tile >= a.n_tiles) return;
    const mg_chunk ck = chunks[chunk_id];
    const int64_t b0 = (int64_t)tile * MG_NCAND;
    const int ncand = (int)((a.B - b0) < MG_NCAND ? (a.B - b0) : MG_NCAND);
    const int stride = a.stride, D = a.D, Dp = a.Dp, L = a.L, nroot = a.nroot;

    // LDS carve-up (all 16-byte aligned: stride % 4 == 0)
    float *lds_c = (float *)smem;                                          // [16][stride]
    float4 *lds_w = (float4 *)(lds_c + (size_t)MG_NCAND * stride);         // [MG_MAX_NT] weights
    float *lds_ro = (float *)(lds_w + MG_MAX_NT);                          // [16][MG_MAX_NT][4] root outputs
    int *lds_m = (int *)(lds_ro + MG_NCAND * MG_MAX_NT * 4);               // [MG_MAX_NT] first tap - imin
    double *lds_root = (double *)(lds_m + MG_MAX_NT);                      // [16][max_wi*nroot+1]
    const int root_stride = a.max_wi * nroot + 1;

    if (tid < ck.nT) {
        lds_w[tid] = w32[ck.t0 + tid];
        lds_m[tid] = i0tab[ck.t0 + tid] - ck.imin;
    }

    // latent tile as MFMA B fragments: lane l supplies B[k = 4*kk + (l >> 4)][n = l & 15]
    const int cl = lane & 15, g = lane >> 4;
    float sfrag[KK];
    double s64frag[KK];
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
        const int k = 4 * kk + g;
        const double v = (cl < ncand && k < L) ? mg_load_lat<LAT_F64>(lat, (b0 + cl) * a.ld + k) : 0.0;
        s64frag[kk] = v;
        sfrag[kk] = (float)v;
    }

    // ---- stage 1a: f32 MFMA over the window's 16-row tiles, two tiles in flight, next two prefetched
    if (!(a.debug & 1)) {
        constexpr int NW = MG_BLOCK / 64;
        const float2 *ep = (const float2 *)Epack;
        float2 a0[KK / 2], a1[KK / 2], n0[KK / 2], n1[KK / 2];
        f32x4 m0, m1, mn0, mn1;
        auto load_tile = [&](int t, float2(&fr)[KK / 2], f32x4 &cin) {
            const int tc = t < ck.ntiles ? t : ck.ntiles - 1;   // clamp: redundant but in-bounds
            const float2 *p = ep + ((size_t)(ck.rt0 + tc) * (KK / 2)) * 64 + lane;
#pragma unroll
            for (int q = 0; q < KK / 2; q++) fr[q] = p[q * 64];
            cin = *(const f32x4 *)(mean32 + (size_t)(ck.rt0 + tc) * 16 + 4 * g);
        };
        int t = wave;
        if (t < ck.ntiles) { load_tile(t, a0, m0); load_tile(t + NW, a1, m1); }
        while (t < ck.ntiles) {
            const int tn = t + 2 * NW;
            if (tn < ck.ntiles) { load_tile(tn, n0, mn0); load_tile(tn + NW, n1, mn1); }
            f32x4 acc0 = m0, acc1 = m1;
#pragma unroll
            for (int q = 0; q < KK / 2; q++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].x, sfrag[2 * q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].x, sfrag[2 * q], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].y, sfrag[2 * q + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].y, sfrag[2 * q + 1], acc1, 0, 0, 0);
            }
            // D[row = 4g + reg][col = cl]: four consecutive padded rows of candidate cl
            *(f32x4 *)&lds_c[cl * stride + t * 16 + 4 * g] = acc0;
            if (t + NW < ck.ntiles) *(f32x4 *)&lds_c[cl * stride + (t + NW) * 16 + 4 * g] = acc1;
#pragma unroll
            for (int q = 0; q < KK / 2; q++) { a0[q] = n0[q]; a1[q] = n1[q]; }
            m0 = mn0; m1 = mn1;
            t = tn;
        }
    }
    // ---- stage 1b: root-translation rows in float64 (rows rr = i*nroot + d) ---------------
    for (int t = wave; t < ck.nrt && !(a.debug & 1); t += MG_BLOCK / 64) {
        const double *p = Erpack + ((size_t)(ck.rrt0 + t) * KK) * 64 + lane;
        const int row0 = (ck.rrt0 + t) * 16;
        // v_mfma_f64_16x16x4_f64 C/D: col = lane & 15, row = (lane >> 4) + 4*reg
        f64x4 acc;
        acc[0] = meanroot[row0 + g];
        acc[1] = meanroot[row0 + g + 4];
        acc[2] = meanroot[row0 + g + 8];
        acc[3] = meanroot[row0 + g + 12];
#pragma unroll
        for (int kk = 0; kk < KK; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(p[kk * 64], s64frag[kk], acc, 0, 0, 0);
        const int lr0 = row0 + g - ck.imin * nroot;   // local root row of reg 0
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int lr = lr0 + 4 * r;
            if (lr >= 0 && lr < ck.wi * nroot) lds_root[cl * root_stride + lr] = acc[r];
        }
    }
    __syncthreads();

    // ---- stage 1c: root channels' spline taps in float64 -> float32 outputs in LDS -------------
    {
        const int n_items = ck.nT * MG_NCAND * nroot;
        for (int it = tid; it < n_items; it += MG_BLOCK) {
            const int c = it & 15;
            const int rest = it >> 4;
            const int fl = rest / nroot, d = rest - fl * nroot;
            const double *wq = w64 + 4 * (size_t)(ck.t0 + fl);
            const double *q = lds_root + c * root_stride + lds_m[fl] * nroot + d;
            double v = wq[0] * q[0];
            v = fma(wq[1], q[nroot], v);
            v = fma(wq[2], q[2 * nroot], v);
            v = fma(wq[3], q[3 * nroot], v);
            lds_ro[(c * MG_MAX_NT + fl) * 4 + d] = (float)v;
        }
    }
    __syncthreads();

    // ---- stage 2: quad-row sweep ----------------------------------------------------------------
    if (a.debug & 2) return;
    const int64_t TD = (int64_t)a.T * D;
    const int nq = Dp >> 2;                       // quads per time sample
    const int rpi = 64 / nq;                      // time samples per wave instruction
    const int fsub = lane / nq, q4 = (lane - fsub * nq) * 4;
    const bool lane_on = lane < rpi * nq;
    const int col0 = ck.imin * Dp - ck.rt0 * 16;  // LDS column of (imin, d = 0)
    const int nvalid = D - q4 < 4 ? D - q4 : 4;   // channels of this quad that exist (last quad of a row)
    for (int c = wave; c < ncand; c += MG_BLOCK / 64) {
        const float *cimg = lds_c + c * stride + col0 + q4;
        float *orow = out + (size_t)(b0 + c) * TD + (size_t)ck.t0 * D + q4;
        for (int f0 = 0; f0 < ck.nT; f0 += rpi) {
            const int fl = f0 + fsub;
            if (lane_on && fl < ck.nT) {
                const float4 w = lds_w[fl];
                const float *tp = cimg + lds_m[fl] * Dp;
                const f32x4 t0 = *(const f32x4 *)tp;
                const f32x4 t1 = *(const f32x4 *)(tp + Dp);
                const f32x4 t2 = *(const f32x4 *)(tp + 2 * Dp);
                const f32x4 t3 = *(const f32x4 *)(tp + 3 * Dp);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float x = w.x * t0[e];
                    x = fmaf(w.y, t1[e], x);
                    x = fmaf(w.z, t2[e], x);
                    x = fmaf(w.w, t3[e], x);
                    v[e] = x;
                }
                if (q4 == 0) {   // root channels come from the float64 path
                    const f32x4 r = *(const f32x4 *)&lds_ro[(c * MG_MAX_NT + fl) * 4];
#pragma unroll
                    for (int e = 0; e < 3; e++)
                        if (e < nroot) v[e] = r[e];
                }
                float *op = orow + (size_t)fl * D;
                if (nvalid == 4) {
                    *(f32x4u *)op = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 3; e++)
                        if (e < nvalid) op[e] = v[e];
                }
            }
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
template <int KK>
static int mg_launch_mfma_kk(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, const mg_frames_args &a,
                             bool lat_f64, int lds, int grid) {
    hipStream_t st = p->ctx->stream;
    if (lat_f64)
        hipLaunchKernelGGL((mg_frames_mfma_kernel<KK, true>), dim3(grid), dim3(MG_BLOCK), lds, st, p->d_Epack, p->d_mean32,
                           p->d_Erpack, p->d_meanroot, lat, g->d_i0, (const float4 *)g->d_w32, g->d_w, g->d_chunks, out, a);
    else
        hipLaunchKernelGGL((mg_frames_mfma_kernel<KK, false>), dim3(grid), dim3(MG_BLOCK), lds, st, p->d_Epack, p->d_mean32,
                           p->d_Erpack, p->d_meanroot, lat, g->d_i0, (const float4 *)g->d_w32, g->d_w, g->d_chunks, out, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

template <int KK>
static int mg_set_attr_kk() {
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_mfma_kernel<KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_mfma_kernel<KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return MG_OK;
}

int mg_setup_kernel_attributes(mg_context *) {
    int rc;
    if ((rc = mg_set_attr_kk<2>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<4>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<6>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<8>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<10>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<12>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<14>()) != MG_OK) return rc;
    if ((rc = mg_set_attr_kk<16>()) != MG_OK) return rc;
    return MG_OK;
}

int mg_launch_frames_mfma(mg_primitive *p, const mg_time_grid *g, const void *lat, int ldt, int64_t B, int64_t ld, float *out) {
    mg_frames_args a;
    a.B = B; a.ld = ld; a.T = g->T; a.D = p->D; a.Dp = p->Dp; a.L = p->L; a.nroot = p->nroot;
    a.n_chunks = g->n_chunks; a.stride = g->stride; a.max_wi = g->max_wi;
    {
        static const int dbg = getenv("MG_DEBUG_FLAGS") ? atoi(getenv("MG_DEBUG_FLAGS")) : 0;
        a.debug = dbg;
    }
    if ((int64_t)g->T * p->D * MG_NCAND * 4 >= ((int64_t)1 << 31)) {
        mg_set_error("mg_back_project_frames: n_times * n_dim too large for the MFMA path");
        return MG_ERR_UNSUPPORTED;
    }
    int64_t n_tiles = (B + MG_NCAND - 1) / MG_NCAND;
    int64_t groups = (n_tiles + 7) / 8;
    int64_t grid = groups * 8 * g->n_chunks;
    if (grid > 0x7fffffff) { mg_set_error("mg_back_project_frames: batch too large for one launch"); return MG_ERR_UNSUPPORTED; }
    a.n_tiles = (int32_t)n_tiles;
    const bool lf = (ldt == MG_F64);
    switch (p->KK) {
        case 2: return mg_launch_mfma_kk<2>(p, g, lat, out, a, lf, g->lds_bytes, (int)grid);
        case 4: return mg_launch_mfma_kk<4>(p, g, lat, out, a, lf, g->lds_bytes, (int)grid);
        case 6: return mg_launch_mfma_kk<6>(p, g, lat, out, a, lf, g->lds_bytes, (int)grid);
        case 8: return mg_launch_mfma_kk<8>(p, g, lat, out, a, lf, g->lds_bytes, (int)grid);
        case 10: return mg_launch_mfma_kk<10>(p, g, lat, out, a, lf, g->lds_bytes, (int)grid);
        case 12: return mg_launch_mfma_kk<12>(p, g, lat, out, a, lf, g->lds_bytes, (int)grid);
        case 14: return mg_launch_mfma_kk<14>(p, g, lat, out, a, lf, g->lds_bytes, (int)grid);
        case 16: return mg_launch_mfma_kk<16>(p, g, lat, out, a, lf, g->lds_bytes, (int)grid);
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

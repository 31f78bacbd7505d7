// Back-projection kernels for gfx950 (MI355X):
//   frames[b][f][d] = sum_j w[f][j] * ( E'[(i0[f]+j) D + d] . s_b )  +  MF[f][d]
// replacing MotionPrimitive.back_project(s, False).get_motion_vector()
// (reference morphablegraphs/motion_model/motion_primitive.py:206-256 and
//  morphablegraphs/motion_model/motion_spline.py:71-92).
//
// f32 arithmetic contract (bit-exact CPU model: oracle/mg_oracle.c, *_f32model):
//   channels d >= nroot : c[r] = fmaf chain over k ascending from 0.0f (== the
//       v_mfma_f32_16x16x4_f32 accumulation order), v = w0*c0, fmaf(w1,c1,v), fmaf(w2,..), fmaf(w3,..),
//       out = hi + (lo + v) with hi/lo the float32 split of the float64 mean frame;
//   channels d <  nroot : float64 fma chains, out = (float)(MF + v64).
#include "mg_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct mg_frames_args {
    const float *Epack;     // [RT][KK/2][64][2]
    const double *Eroot;    // [NB*nroot][L]
    const void *lat;        // (B, ld) f32 or f64
    const int32_t *i0;      // (T)
    const double *w;        // (T,4)
    const double *mf;       // (T,D)
    const mg_chunk *chunks;
    float *out;             // (B,T,D)
    int64_t B, ld;
    int32_t T, D, L, nroot, n_chunks, n_tiles, stride, max_wi, lat_f64;
};

template <bool F64>
__device__ __forceinline__ double mg_load_lat(const void *lat, int64_t idx) {
    if (F64) return ((const double *)lat)[idx];
    return (double)((const float *)lat)[idx];
}

// One workgroup = 16 candidates x one time chunk.
//   stage 1: the chunk's coefficient window (<= 8 basis functions x D channels) for the 16
//            candidates by v_mfma_f32_16x16x4_f32 (A = E' fragments streamed from L2,
//            B = the latent tile held in registers), accumulators -> LDS image [cand][row];
//            root-translation rows in float64 on the VALU -> LDS.
//   stage 2: each thread owns output elements (f, d) of the chunk and walks the 16 candidates:
//            4 LDS taps, 3 fma, 2 adds, one coalesced dword store per candidate.
template <int KK, bool LAT_F64>
__global__ __launch_bounds__(MG_BLOCK) void mg_frames_mfma_kernel(mg_frames_args a) {
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
    const mg_chunk ck = a.chunks[chunk_id];
    const int64_t b0 = (int64_t)tile * MG_NCAND;
    const int ncand = (int)((a.B - b0) < MG_NCAND ? (a.B - b0) : MG_NCAND);
    const int stride = a.stride, D = a.D, L = a.L, nroot = a.nroot;

    float *lds_c = (float *)smem;                                          // [16][stride]
    double *lds_root = (double *)(smem + (size_t)MG_NCAND * stride * 4);   // [16][max_wi*nroot+1]
    const int root_stride = a.max_wi * nroot + 1;
    double *lds_s = lds_root + (size_t)MG_NCAND * root_stride;             // [16][L+1]
    const int s_stride = L + 1;

    // latent tile: float64 copy in LDS for the root rows, float32 B fragments in registers
    for (int e = tid; e < MG_NCAND * L; e += MG_BLOCK) {
        int c = e / L, k = e - c * L;
        double v = (c < ncand) ? mg_load_lat<LAT_F64>(a.lat, (b0 + c) * a.ld + k) : 0.0;
        lds_s[c * s_stride + k] = v;
    }
    float sfrag[KK];
    {
        const int c = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            int k = 4 * kk + kq;
            sfrag[kk] = (c < ncand && k < L) ? (float)mg_load_lat<LAT_F64>(a.lat, (b0 + c) * a.ld + k) : 0.0f;
        }
    }
    __syncthreads();

    // ---- stage 1a: MFMA contraction over the window's 16-row tiles ---------------------
    {
        const int c = lane & 15, g = lane >> 4;
        for (int t = wave; t < ck.ntiles; t += 2 * (MG_BLOCK / 64)) {
            const int t2 = t + MG_BLOCK / 64;
            const bool has2 = t2 < ck.ntiles;   // wave-uniform
            const float2 *ap0 = (const float2 *)a.Epack + ((size_t)(ck.rt0 + t) * (KK / 2)) * 64 + lane;
            const float2 *ap1 = (const float2 *)a.Epack + ((size_t)(ck.rt0 + (has2 ? t2 : t)) * (KK / 2)) * 64 + lane;
            float2 a0[KK / 2], a1[KK / 2];
#pragma unroll
            for (int q = 0; q < KK / 2; q++) { a0[q] = ap0[q * 64]; a1[q] = ap1[q * 64]; }
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KK / 2; q++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].x, sfrag[2 * q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].x, sfrag[2 * q], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].y, sfrag[2 * q + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].y, sfrag[2 * q + 1], acc1, 0, 0, 0);
            }
            // D[row = 4g + reg][col = c]: four consecutive rows of candidate c
            *(f32x4 *)&lds_c[c * stride + t * 16 + 4 * g] = acc0;
            if (has2) *(f32x4 *)&lds_c[c * stride + t2 * 16 + 4 * g] = acc1;
        }
    }
    // ---- stage 1b: root-translation rows in float64 ------------------------------------
    {
        const int npairs = ck.wi * nroot * MG_NCAND;
        for (int pidx = tid; pidx < npairs; pidx += MG_BLOCK) {
            const int c = pidx & 15, rr = pidx >> 4;   // rr = i_local * nroot + d
            const double *er = a.Eroot + ((size_t)ck.imin * nroot + rr) * L;
            const double *sv = lds_s + c * s_stride;
            double acc = 0.0;
            for (int k = 0; k < L; k++) acc = fma(er[k], sv[k], acc);
            lds_root[c * root_stride + rr] = acc;
        }
    }
    __syncthreads();

    // ---- stage 2: spline taps + mean frame, coalesced stores -----------------------------
    const int64_t TD = (int64_t)a.T * D;
    const int n_out = ck.nT * D;
    const int base_local = ck.rt0 * 16;   // global row of LDS column 0
    for (int o = tid; o < n_out; o += MG_BLOCK) {
        const int fl = o / D;
        const int d = o - fl * D;
        if (d < nroot) continue;
        const int f = ck.t0 + fl;
        const int i0v = a.i0[f];
        const double4 wd = *(const double4 *)(a.w + 4 * (size_t)f);
        const float w0 = (float)wd.x, w1 = (float)wd.y, w2 = (float)wd.z, w3 = (float)wd.w;
        const double mfd = a.mf[(size_t)f * D + d];
        const float hi = (float)mfd;
        const float lo = (float)(mfd - (double)hi);
        const float *cp = lds_c + (i0v * D + d - base_local);
        float *op = a.out + (size_t)b0 * TD + (size_t)f * D + d;
#pragma unroll 4
        for (int c = 0; c < ncand; c++) {
            const float *q = cp + c * stride;
            float v = w0 * q[0];
            v = fmaf(w1, q[D], v);
            v = fmaf(w2, q[2 * D], v);
            v = fmaf(w3, q[3 * D], v);
            op[(size_t)c * TD] = hi + (lo + v);
        }
    }
    // root channels: (f, d < nroot, cand)
    {
        const int n_items = ck.nT * nroot * MG_NCAND;
        for (int it = tid; it < n_items; it += MG_BLOCK) {
            const int c = it & 15;
            const int rest = it >> 4;
            const int fl = rest / nroot, d = rest - fl * nroot;
            if (c >= ncand) continue;
            const int f = ck.t0 + fl;
            const int i0v = a.i0[f];
            const double *wq = a.w + 4 * (size_t)f;
            const double *q = lds_root + c * root_stride + (i0v - ck.imin) * nroot + d;
            double v = wq[0] * q[0];
            v = fma(wq[1], q[nroot], v);
            v = fma(wq[2], q[2 * nroot], v);
            v = fma(wq[3], q[3 * nroot], v);
            a.out[(size_t)(b0 + c) * TD + (size_t)f * D + d] = (float)(a.mf[(size_t)f * D + d] + v);
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
    const void *lat;
    const int32_t *i0;
    const double *w;
    const double *mf;
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
        const double mfd = a.mf[(size_t)f * a.D + d];
        const int r0 = i0v * a.D + d;
        if (OUT_F64 || d < a.nroot) {
            double c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double *e = a.Et64 + (r0 + j * a.D);
                double acc = 0.0;
                for (int k = 0; k < a.L; k++) acc = fma(e[(size_t)k * a.R], mg_load_lat<LAT_F64>(a.lat, b * a.ld + k), acc);
                c[j] = acc;
            }
            double v = wq[0] * c[0];
            v = fma(wq[1], c[1], v);
            v = fma(wq[2], c[2], v);
            v = fma(wq[3], c[3], v);
            if (OUT_F64) ((double *)a.out)[idx] = mfd + v;
            else ((float *)a.out)[idx] = (float)(mfd + v);
        } else {
            float c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float *e = a.Et32 + (r0 + j * a.D);
                float acc = 0.0f;
                for (int k = 0; k < a.L; k++) acc = fmaf(e[(size_t)k * a.R], (float)mg_load_lat<LAT_F64>(a.lat, b * a.ld + k), acc);
                c[j] = acc;
            }
            const float hi = (float)mfd, lo = (float)(mfd - (double)hi);
            float v = (float)wq[0] * c[0];
            v = fmaf((float)wq[1], c[1], v);
            v = fmaf((float)wq[2], c[2], v);
            v = fmaf((float)wq[3], c[3], v);
            ((float *)a.out)[idx] = hi + (lo + v);
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
static int mg_launch_mfma_kk(mg_primitive *p, const mg_frames_args &a, int lds, int grid) {
    hipStream_t st = p->ctx->stream;
    if (a.lat_f64) hipLaunchKernelGGL((mg_frames_mfma_kernel<KK, true>), dim3(grid), dim3(MG_BLOCK), lds, st, a);
    else hipLaunchKernelGGL((mg_frames_mfma_kernel<KK, false>), dim3(grid), dim3(MG_BLOCK), lds, st, a);
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
    a.Epack = p->d_Epack; a.Eroot = p->d_Eroot; a.lat = lat;
    a.i0 = g->d_i0; a.w = g->d_w; a.mf = g->d_mf; a.chunks = g->d_chunks; a.out = out;
    a.B = B; a.ld = ld; a.T = g->T; a.D = p->D; a.L = p->L; a.nroot = p->nroot;
    a.n_chunks = g->n_chunks; a.stride = g->stride; a.max_wi = g->max_wi; a.lat_f64 = (ldt == MG_F64);
    int64_t n_tiles = (B + MG_NCAND - 1) / MG_NCAND;
    int64_t groups = (n_tiles + 7) / 8;
    int64_t grid = groups * 8 * g->n_chunks;
    if (grid > 0x7fffffff) { mg_set_error("mg_back_project_frames: batch too large for one launch"); return MG_ERR_UNSUPPORTED; }
    a.n_tiles = (int32_t)n_tiles;
    switch (p->KK) {
        case 2: return mg_launch_mfma_kk<2>(p, a, g->lds_bytes, (int)grid);
        case 4: return mg_launch_mfma_kk<4>(p, a, g->lds_bytes, (int)grid);
        case 6: return mg_launch_mfma_kk<6>(p, a, g->lds_bytes, (int)grid);
        case 8: return mg_launch_mfma_kk<8>(p, a, g->lds_bytes, (int)grid);
        case 10: return mg_launch_mfma_kk<10>(p, a, g->lds_bytes, (int)grid);
        case 12: return mg_launch_mfma_kk<12>(p, a, g->lds_bytes, (int)grid);
        case 14: return mg_launch_mfma_kk<14>(p, a, g->lds_bytes, (int)grid);
        case 16: return mg_launch_mfma_kk<16>(p, a, g->lds_bytes, (int)grid);
        default: mg_set_error("mg_back_project_frames: MFMA path needs n_components <= 64"); return MG_ERR_UNSUPPORTED;
    }
}

int mg_launch_frames_direct(mg_primitive *p, const mg_time_grid *g, const void *lat, int ldt, int64_t B, int64_t ld, void *out, bool out_f64) {
    mg_direct_args a;
    a.Et32 = p->d_Et32; a.Et64 = p->d_Et64; a.lat = lat; a.i0 = g->d_i0; a.w = g->d_w; a.mf = g->d_mf; a.out = out;
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

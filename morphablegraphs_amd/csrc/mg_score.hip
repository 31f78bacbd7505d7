// Fused candidate scoring for gfx950 (MI355X), float64 arithmetic, no frames materialised:
//   error_b = sum_c weight_c * constraint_c( root pose of candidate b at keyframe t_c )
// replacing the per-sample loop of evaluate_samples_using_constraints
// (reference morphablegraphs/motion_generator/motion_primitive_generator.py:230-261) over
// MotionPrimitiveConstraints.evaluate (reference .../constraints/motion_primitive_constraints.py:100-122)
// for root-joint position / 2-D direction constraints, and its first-minimum argmin.
#include "mg_internal.h"
#include <cstring>
#include "mg_gmm_device.h"

#include "mg_score_device.h"

// VALU kernel (fallback for > 64 latent components): one workgroup = 64 candidates (a lane each) x 4 waves that deal
// the constraints round-robin.  The latent tile is staged in LDS ([64][L+1] float64), the fused keyframe matrices are
// wave-uniform (scalar loads); every weighted residual meets in LDS ([n][64]) and lane-owners sum them in constraint
// order (the order MotionPrimitiveConstraints.evaluate adds them in).
#define MG_SC_CANDS 64
#define MG_SC_WAVES 4
template <bool LAT_F64, bool OUT_F64>
__global__ __launch_bounds__(MG_SC_CANDS *MG_SC_WAVES) void mg_score_kernel(mg_score_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = a.L, xs = L + 1;
    double *lds_x = (double *)smem;                       // [64][L+1]
    double *lds_r = lds_x + MG_SC_CANDS * xs;             // [n][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * MG_SC_CANDS;
    const int ncand = (int)((a.B - b0) < MG_SC_CANDS ? (a.B - b0) : MG_SC_CANDS);
    for (int e = tid; e < MG_SC_CANDS * L; e += MG_SC_CANDS * MG_SC_WAVES) {
        int c = e / L, i = e - c * L;
        double v = 0.0;
        if (c < ncand) v = LAT_F64 ? ((const double *)a.lat)[(b0 + c) * a.ld + i] : (double)((const float *)a.lat)[(b0 + c) * a.ld + i];
        lds_x[c * xs + i] = v;
    }
    __syncthreads();
    const double *x = lds_x + lane * xs;
    for (int c = wave; c < a.n; c += MG_SC_WAVES) {
        auto channel = [&](int row) {   // one pose channel of this candidate at the keyframe: fma chain over k from the bias
            const double *wr = a.W + (size_t)row * L;
            double acc = a.bias[row];
            for (int k = 0; k < L; k++) acc = fma(wr[k], x[k], acc);
            return acc;
        };
        lds_r[c * MG_SC_CANDS + lane] = mg_constraint_residual(a, c, channel, b0 + lane);
    }
    __syncthreads();
    if (a.res)   // (n_samples, n) row-major: consecutive threads write consecutive constraints of a candidate
        for (int e = tid; e < ncand * a.n; e += MG_SC_CANDS * MG_SC_WAVES) {
            const int cand = e / a.n, c = e - cand * a.n;
            a.res[(b0 + cand) * a.n + c] = lds_r[c * MG_SC_CANDS + cand];
        }
    if (a.out && tid < ncand) {
        double err = 0.0;
        for (int c = 0; c < a.n; c++) err += lds_r[c * MG_SC_CANDS + tid];
        if (OUT_F64) ((double *)a.out)[b0 + tid] = err;
        else ((float *)a.out)[b0 + tid] = (float)err;
    }
}

// MFMA kernel (n_components <= 64): every pose channel the constraints need is one row of the fused keyframe
// matrices, so all channels of 16 candidates are ONE small GEMM, X (16 x L) . W^T (L x rows) + bias, on the float64
// matrix pipe: per 16-row tile KK chained v_mfma_f64_16x16x4_f64 with C-in = bias -- the same k-ordered fma chain
// as the VALU kernel's dot products, hence the same bits.  A wave owns a 16-candidate tile: channels -> LDS
// ([16][rows+1]), then its lanes take (candidate, constraint) pairs, compute the residuals (FK chains included)
// from LDS, and lanes < 16 sum a candidate's residuals in constraint order.
template <int KK, bool LAT_F64, bool OUT_F64>
__global__ __launch_bounds__(256) void mg_score_mfma_kernel(mg_score_args a, const double *__restrict__ Wpack,   // [RT][KK][64]
                                                           const double *__restrict__ bpad,                    // [RT*16]
                                                           const int RT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;
    const int vs = RT * 16 + 1;                                          // padded channel row of a candidate
    double *vals = (double *)smem + (size_t)wave * (16 * vs + a.n * 16);  // [16][vs]
    double *resid = vals + 16 * vs;                                      // [n][16]
    const int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
    if (b0 >= a.B) return;                                               // no workgroup-wide barrier below
    const int ncand = (int)((a.B - b0) < 16 ? (a.B - b0) : 16);
    typename mg_gmm_xt<LAT_F64>::type xf[KK];
    mg_gmm_load_x<KK, LAT_F64>(xf, a.lat, b0, ncand, a.ld, a.L, cl, g);
    for (int rt = 0; rt < RT; rt++) {
        const double *wp = Wpack + ((size_t)rt * KK) * 64 + lane;
        const double c0 = bpad[rt * 16 + cl];
        mg_f64x4 acc = {c0, c0, c0, c0};
#pragma unroll
        for (int kk = 0; kk < KK; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xf[kk], wp[kk * 64], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) vals[(g + 4 * r) * vs + rt * 16 + cl] = acc[r];   // C layout: col = row index, row = candidate
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its own LDS writes are visible to its reads in order
    for (int e = lane; e < 16 * a.n; e += 64) {
        const int cand = e & 15, c = e >> 4;
        const double *v = vals + cand * vs;
        resid[c * 16 + cand] = mg_constraint_residual(a, c, [&](int row) { return v[row]; }, b0 + cand);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (a.res)
        for (int e = lane; e < ncand * a.n; e += 64) {
            const int cand = e / a.n, c = e - cand * a.n;
            a.res[(b0 + cand) * a.n + c] = resid[c * 16 + cand];
        }
    if (a.out && lane < ncand) {
        double err = 0.0;
        for (int c = 0; c < a.n; c++) err += resid[c * 16 + lane];
        if (OUT_F64) ((double *)a.out)[b0 + lane] = err;
        else ((float *)a.out)[b0 + lane] = (float)err;
    }
}

// The same for large batches: a wave owns FOUR tiles, 64 candidates.  The matrix part is the same chains (four tiles' latents
// requested up front); in the residual part a LANE is a candidate and walks the constraints in order -- every lane busy and all
// lanes in the same constraint's code at the same time, where the kernel above has (candidate, constraint) pairs on the lanes: with
// two constraints of different kinds half its lanes idle and the two kinds' float64 arithmetic (square roots, arc tangents) runs one
// after the other at a quarter of the wave each.  Same residual function, same order of the sum: the same bits.
template <int KK, bool LAT_F64, bool OUT_F64>
__global__ __launch_bounds__(256) void mg_score_mfma64_kernel(mg_score_args a, const double *__restrict__ Wpack, const double *__restrict__ bpad,
                                                             const int RT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;
    const int vs = RT * 16 + 1;
    double *vals = (double *)smem + (size_t)wave * (64 * vs);            // [64][vs]
    const int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (b0 >= a.B) return;
    const int ncand = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
    typename mg_gmm_xt<LAT_F64>::type xf[4][KK];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int nt = ncand - 16 * t;                                   // candidates of tile t (a tile past the end re-reads the last row)
        mg_gmm_load_x<KK, LAT_F64>(xf[t], a.lat, nt > 0 ? b0 + 16 * t : b0, nt > 0 ? (nt < 16 ? nt : 16) : 1, a.ld, a.L, cl, g);
    }
    for (int rt = 0; rt < RT; rt++) {
        const double *wp = Wpack + ((size_t)rt * KK) * 64 + lane;
        const double c0 = bpad[rt * 16 + cl];
        mg_f64x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = {c0, c0, c0, c0};
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            const double w = wp[kk * 64];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xf[t][kk], w, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) vals[(16 * t + g + 4 * r) * vs + rt * 16 + cl] = acc[t][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < ncand) {
        const double *v = vals + lane * vs;
        double err = 0.0;
        for (int c = 0; c < a.n; c++) {
            const double r = mg_constraint_residual(a, c, [&](int row) { return v[row]; }, b0 + lane);
            if (a.res) a.res[(b0 + lane) * a.n + c] = r;
            err += r;
        }
        if (a.out) {
            if (OUT_F64) ((double *)a.out)[b0 + lane] = err;
            else ((float *)a.out)[b0 + lane] = (float)err;
        }
    }
}

#define MG_SCORE_WIDE_MIN_B 49152   // measured crossover (tools/probes/score_kernel_ab.py): below, the tile kernel's eight waves per SIMD hide more latency than its idle lanes cost

template <int KK>
static int mg_launch_score_mfma_kk(mg_primitive *p, const mg_constraint_set *cs, const mg_score_args &a, bool lf, bool of) {
    hipStream_t st = p->ctx->stream;
    const int kernel_opt = p->ctx->opt[MG_OPT_SCORE_KERNEL];   // 0: by batch size, 1: 16-candidate waves, 2: 64-candidate waves
    const bool wide = kernel_opt == 2 || (kernel_opt == 0 && a.B >= MG_SCORE_WIDE_MIN_B);
    if (wide) {
        const size_t lds = (size_t)4 * 64 * (cs->RT * 16 + 1) * 8;
        if (lds <= 64 * 1024) {
            const int64_t grid = (a.B + 255) / 256;
            if (grid > 0x7fffffff) return MG_ERR_UNSUPPORTED;
            if (lf && of) hipLaunchKernelGGL((mg_score_mfma64_kernel<KK, true, true>), dim3((int)grid), dim3(256), lds, st, a, cs->d_Wpack, cs->d_bpad, cs->RT);
            else if (lf) hipLaunchKernelGGL((mg_score_mfma64_kernel<KK, true, false>), dim3((int)grid), dim3(256), lds, st, a, cs->d_Wpack, cs->d_bpad, cs->RT);
            else if (of) hipLaunchKernelGGL((mg_score_mfma64_kernel<KK, false, true>), dim3((int)grid), dim3(256), lds, st, a, cs->d_Wpack, cs->d_bpad, cs->RT);
            else hipLaunchKernelGGL((mg_score_mfma64_kernel<KK, false, false>), dim3((int)grid), dim3(256), lds, st, a, cs->d_Wpack, cs->d_bpad, cs->RT);
            MG_HIP_CHECK(hipGetLastError());
            return MG_OK;
        }
    }
    const size_t lds = (size_t)4 * (16 * (cs->RT * 16 + 1) + (size_t)std::max(cs->n, 1) * 16) * 8;
    if (lds > 150 * 1024) return MG_ERR_UNSUPPORTED;
    const int64_t grid = (a.B + 63) / 64;
    if (grid > 0x7fffffff) return MG_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_score_mfma_kernel<KK, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_score_mfma_kernel<KK, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_score_mfma_kernel<KK, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_score_mfma_kernel<KK, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (lf && of) hipLaunchKernelGGL((mg_score_mfma_kernel<KK, true, true>), dim3((int)grid), dim3(256), lds, st, a, cs->d_Wpack, cs->d_bpad, cs->RT);
    else if (lf) hipLaunchKernelGGL((mg_score_mfma_kernel<KK, true, false>), dim3((int)grid), dim3(256), lds, st, a, cs->d_Wpack, cs->d_bpad, cs->RT);
    else if (of) hipLaunchKernelGGL((mg_score_mfma_kernel<KK, false, true>), dim3((int)grid), dim3(256), lds, st, a, cs->d_Wpack, cs->d_bpad, cs->RT);
    else hipLaunchKernelGGL((mg_score_mfma_kernel<KK, false, false>), dim3((int)grid), dim3(256), lds, st, a, cs->d_Wpack, cs->d_bpad, cs->RT);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

int mg_launch_score(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int ldt, int64_t B, int64_t ld, void *out, int odt, double *res, const double *align_cand) {
    mg_score_args a;
    a.res = res; a.align_cand = align_cand;
    a.W = cs->d_W; a.bias = cs->d_bias; a.par = cs->d_par; a.woff = cs->d_woff; a.chain = cs->d_chain; a.choff = cs->d_choff; a.align = cs->d_align; a.pose = cs->d_pose; a.lat = lat; a.out = out; a.B = B; a.ld = ld; a.n = cs->n; a.nch = cs->nch; a.L = p->L;
    const bool lf0 = ldt == MG_F64, of0 = odt == MG_F64;
    if (cs->d_Wpack && !p->ctx->opt[MG_OPT_FORCE_VALU_SCORE]) {
        int rc = MG_ERR_UNSUPPORTED;
        switch (p->KK) {
            case 2: rc = mg_launch_score_mfma_kk<2>(p, cs, a, lf0, of0); break;
            case 4: rc = mg_launch_score_mfma_kk<4>(p, cs, a, lf0, of0); break;
            case 6: rc = mg_launch_score_mfma_kk<6>(p, cs, a, lf0, of0); break;
            case 8: rc = mg_launch_score_mfma_kk<8>(p, cs, a, lf0, of0); break;
            case 10: rc = mg_launch_score_mfma_kk<10>(p, cs, a, lf0, of0); break;
            case 12: rc = mg_launch_score_mfma_kk<12>(p, cs, a, lf0, of0); break;
            case 14: rc = mg_launch_score_mfma_kk<14>(p, cs, a, lf0, of0); break;
            case 16: rc = mg_launch_score_mfma_kk<16>(p, cs, a, lf0, of0); break;
            default: break;
        }
        if (rc != MG_ERR_UNSUPPORTED) return rc;
    }
    size_t lds = ((size_t)MG_SC_CANDS * (p->L + 1) + (size_t)std::max(cs->n, 1) * MG_SC_CANDS) * 8;
    if (lds > 150 * 1024) { mg_set_error("mg_score_constraints: n_components %d x %d constraints too large for LDS", p->L, cs->n); return MG_ERR_UNSUPPORTED; }
    int64_t grid = (B + MG_SC_CANDS - 1) / MG_SC_CANDS;
    if (grid > 0x7fffffff) { mg_set_error("mg_score_constraints: too many samples"); return MG_ERR_UNSUPPORTED; }
    hipStream_t st = p->ctx->stream;
    const bool lf = ldt == MG_F64, of = odt == MG_F64;
    if (lds > 64 * 1024) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_score_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_score_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_score_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_score_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (lf && of) hipLaunchKernelGGL((mg_score_kernel<true, true>), dim3((int)grid), dim3(MG_SC_CANDS * MG_SC_WAVES), lds, st, a);
    else if (lf) hipLaunchKernelGGL((mg_score_kernel<true, false>), dim3((int)grid), dim3(MG_SC_CANDS * MG_SC_WAVES), lds, st, a);
    else if (of) hipLaunchKernelGGL((mg_score_kernel<false, true>), dim3((int)grid), dim3(MG_SC_CANDS * MG_SC_WAVES), lds, st, a);
    else hipLaunchKernelGGL((mg_score_kernel<false, false>), dim3((int)grid), dim3(MG_SC_CANDS * MG_SC_WAVES), lds, st, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// New constraint parameters for an existing set (mg_constraint_set_update): the values travel as kernel arguments,
// so the call needs neither a staging buffer that outlives it nor a synchronisation, and is ordered on the stream
// between the launches that read the old values and those that read the new.
#define MG_SET_PARAMS_MAX 496   // doubles; kernel arguments are limited to 4 KB
struct mg_set_params_args { double v[MG_SET_PARAMS_MAX]; };
__global__ __launch_bounds__(64) void mg_set_params_kernel(mg_set_params_args a, int n_par, int n_align, double *par, double *align) {
    for (int i = threadIdx.x; i < n_par; i += 64) par[i] = a.v[i];
    for (int i = threadIdx.x; i < n_align; i += 64) align[i] = a.v[n_par + i];
}
int mg_launch_set_params(mg_context *ctx, const double *values, int n_par, int n_align, double *d_par, double *d_align) {
    if (n_par + n_align > MG_SET_PARAMS_MAX) {   // more than 60 constraints: plain copies behind a synchronisation
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (n_par) MG_HIP_CHECK(hipMemcpy(d_par, values, (size_t)n_par * 8, hipMemcpyHostToDevice));
        if (n_align) MG_HIP_CHECK(hipMemcpy(d_align, values + n_par, (size_t)n_align * 8, hipMemcpyHostToDevice));
        return MG_OK;
    }
    mg_set_params_args a;
    memcpy(a.v, values, (size_t)(n_par + n_align) * 8);
    hipLaunchKernelGGL(mg_set_params_kernel, dim3(1), dim3(64), 0, ctx->stream, a, n_par, n_align, d_par, d_align);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// -----------------------------------------------------------------------------------------
// First-minimum argmin (reference motion_primitive_generator.py:251-257: `if min_error > error`):
// the smallest value wins, ties go to the smallest index, NaN never wins, (0, +inf) if
// nothing wins.  One workgroup: strided scan, wave shuffle reduction, LDS across waves.
// -----------------------------------------------------------------------------------------
template <bool F64>
__global__ __launch_bounds__(1024) void mg_argmin_kernel(const void *vals, int64_t n, void *out) {
    __shared__ double sv[16];
    __shared__ int64_t si[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double best = INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t i = tid; i < n; i += 1024) {
        double v = F64 ? ((const double *)vals)[i] : (double)((const float *)vals)[i];
        if (v < best) { best = v; bi = i; }   // ascending i per thread: strict '<' keeps the first
    }
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_down(best, off, 64);
        long long oi = __shfl_down((long long)bi, off, 64);
        mg_min_combine(best, bi, ov, (int64_t)oi);
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) mg_min_combine(best, bi, sv[w], si[w]);
        if (bi == INT64_MAX) { bi = 0; best = INFINITY; }
        ((int64_t *)out)[0] = bi;
        ((double *)out)[1] = best;
    }
}

int mg_launch_argmin(mg_context *ctx, const void *v, int dt, int64_t n, void *out_dev) {
    if (dt == MG_F64) hipLaunchKernelGGL((mg_argmin_kernel<true>), dim3(1), dim3(1024), 0, ctx->stream, v, n, out_dev);
    else hipLaunchKernelGGL((mg_argmin_kernel<false>), dim3(1), dim3(1024), 0, ctx->stream, v, n, out_dev);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// result = {int64 index, float64 error} written by the argmin kernel, followed by the winning latent row
__global__ __launch_bounds__(64) void mg_gather_winner_kernel(const void *x, int x_f64, int64_t ld, int L, void *result) {
    const int64_t idx = ((const int64_t *)result)[0];
    double *row = (double *)((char *)result + 16);
    for (int i = threadIdx.x; i < L; i += 64)
        row[i] = x_f64 ? ((const double *)x)[idx * ld + i] : (double)((const float *)x)[idx * ld + i];
}

// the two in one launch (a planner step is a chain of small launches: every one saved counts)
template <bool F64>
__global__ __launch_bounds__(1024) void mg_argmin_gather_kernel(const void *vals, int64_t n, void *out, const void *x, int x_f64, int64_t ld, int L, int64_t index_offset) {
    __shared__ double sv[16];
    __shared__ int64_t si[16];
    __shared__ int64_t winner;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double best = INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t i = tid; i < n; i += 1024) {
        double v = F64 ? ((const double *)vals)[i] : (double)((const float *)vals)[i];
        if (v < best) { best = v; bi = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_down(best, off, 64);
        long long oi = __shfl_down((long long)bi, off, 64);
        mg_min_combine(best, bi, ov, (int64_t)oi);
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) mg_min_combine(best, bi, sv[w], si[w]);
        if (bi == INT64_MAX) { bi = 0; best = INFINITY; }
        ((int64_t *)out)[0] = bi + index_offset;   // a rank's block of a sharded step reports the global row
        ((double *)out)[1] = best;
        winner = bi;
    }
    __syncthreads();
    const int64_t idx = winner;
    double *row = (double *)((char *)out + 16);
    for (int i = tid; i < L; i += 1024)
        row[i] = x_f64 ? ((const double *)x)[idx * ld + i] : (double)((const float *)x)[idx * ld + i];
}
int mg_launch_argmin_gather(mg_context *ctx, const void *v, int dt, int64_t n, void *result_dev, const void *x, int xdt, int64_t ld, int L, int64_t index_offset) {
    if (dt == MG_F64) hipLaunchKernelGGL((mg_argmin_gather_kernel<true>), dim3(1), dim3(1024), 0, ctx->stream, v, n, result_dev, x, xdt == MG_F64 ? 1 : 0, ld, L, index_offset);
    else hipLaunchKernelGGL((mg_argmin_gather_kernel<false>), dim3(1), dim3(1024), 0, ctx->stream, v, n, result_dev, x, xdt == MG_F64 ? 1 : 0, ld, L, index_offset);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

int mg_launch_gather_winner(mg_context *ctx, const void *x, int xdt, int64_t ld, int L, void *result_dev) {
    hipLaunchKernelGGL(mg_gather_winner_kernel, dim3(1), dim3(64), 0, ctx->stream, x, xdt == MG_F64 ? 1 : 0, ld, L, result_dev);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}



// Global joint positions of whole frames by forward kinematics: out[n][j] = position of joints[j] in frame n (the inner loop
// of map_motions_to_euclidean_space, reference space_partitioning/features.py:133-153, where anim_utils'
// skeleton.nodes[j].get_global_position(frame) runs once per sample, frame and joint; PARITY UNPINNED like every FK here).
// table: per output joint 1 + 4 * MG_MAX_CHAIN doubles = chain length m, then per link (quaternion channel or -1, offset xyz).
__global__ __launch_bounds__(256) void mg_joint_positions_kernel(const double *__restrict__ frames, const double *__restrict__ table, int64_t N,
                                                                 int D, int J, double *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * J) return;
    const int64_t n = idx / J;
    const int j = (int)(idx - n * J);
    const double *f = frames + n * D;
    const double *rec = table + (size_t)j * (1 + 4 * MG_MAX_CHAIN);
    const int m = (int)rec[0];
    double p0 = f[0], p1 = f[1], p2 = f[2];
    double aw = 1.0, ax = 0.0, ay = 0.0, az = 0.0;
    for (int k = 0; k < m; k++) {
        const int ch = (int)rec[1 + 4 * k];
        if (ch >= 0) {
            double qw = f[ch], qx = f[ch + 1], qy = f[ch + 2], qz = f[ch + 3];
            const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
            qw *= inv; qx *= inv; qy *= inv; qz *= inv;
            const double nw = aw * qw - ax * qx - ay * qy - az * qz, nx = aw * qx + ax * qw + ay * qz - az * qy;
            const double ny = aw * qy - ax * qz + ay * qw + az * qx, nz = aw * qz + ax * qy - ay * qx + az * qw;
            aw = nw; ax = nx; ay = ny; az = nz;
        }
        const double ox = rec[2 + 4 * k], oy = rec[3 + 4 * k], oz = rec[4 + 4 * k];
        const double cx = ay * oz - az * oy, cy = az * ox - ax * oz, cz = ax * oy - ay * ox;
        const double dx = ay * cz - az * cy, dy = az * cx - ax * cz, dz = ax * cy - ay * cx;
        p0 += ox + 2.0 * (aw * cx + dx);
        p1 += oy + 2.0 * (aw * cy + dy);
        p2 += oz + 2.0 * (aw * cz + dz);
    }
    out[idx * 3] = p0; out[idx * 3 + 1] = p1; out[idx * 3 + 2] = p2;
}

int mg_launch_joint_positions(mg_context *ctx, const double *frames, const double *table, int64_t N, int D, int J, double *out) {
    const int64_t total = N * J;
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(mg_joint_positions_kernel, dim3(grid), dim3(256), 0, ctx->stream, frames, table, N, D, J, out);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

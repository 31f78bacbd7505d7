// The direct frames kernel (one thread per output element) and the spline evaluation from explicit coefficients (gfx950).
#include "mg_frames_common.h"

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
    const float *rootm;   // (T, 8): {Mhi[0..2], Mlo[0..2], 0, 0}, the mean/delta split of the root channels
    void *out;
    int64_t B, ld;
    int32_t T, D, L, R, nroot;
    int32_t split;        // float32 output: root channels by the mean/delta split instead of the float64 pipeline
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
        if (!OUT_F64 && d < a.nroot && a.split) {
            float c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float *e = a.Et32 + (r0 + j * a.D);
                float acc = 0.0f;
                for (int k = 0; k < a.L; k++) acc = fmaf(e[(size_t)k * a.R], (float)mg_load_lat<LAT_F64>(a.lat, b * a.ld + k), acc);
                c[j] = acc;
            }
            float v = (float)wq[0] * c[0];
            v = fmaf((float)wq[1], c[1], v);
            v = fmaf((float)wq[2], c[2], v);
            v = fmaf((float)wq[3], c[3], v);
            const float *m = a.rootm + 8 * (size_t)f;
            ((float *)a.out)[idx] = m[d] + (m[3 + d] + v);
        } else if (OUT_F64 || d < a.nroot) {
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


int mg_launch_frames_direct(mg_primitive *p, const mg_time_grid *g, const void *lat, int ldt, int64_t B, int64_t ld, void *out, bool out_f64) {
    mg_direct_args a;
    a.Et32 = p->d_Et32; a.Et64 = p->d_Et64; a.mean = p->d_mean; a.lat = lat; a.i0 = g->d_i0; a.w = g->d_w; a.out = out;
    a.rootm = g->d_rootm; a.split = (!out_f64 && mg_frames_root_split(p)) ? 1 : 0;
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

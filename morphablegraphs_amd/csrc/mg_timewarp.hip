// Time-warped synthesis for a batch (gfx950 / MI355X), float64.
//
// The reference turns a finished graph walk into the motion it returns with back_project(s, use_time_parameters=True)
// (motion_generator/graph_walk.py:154-176): the latent vector's time part gamma gives the canonical time function t(t')
// (motion_model/motion_primitive.py:289-302, on the device since round 2: mg_time_function_kernel), its INVERSE t'(t) sampled at the
// integer sample times is the spline's time function (motion_primitive.py:304-319: scipy.interpolate.splrep through the points
// (t(t'), t'), i.e. FITPACK's interpolating cubic with knots at the data points but the second and the second-to-last one -- the
// not-a-knot cubic -- evaluated by splev at linspace(1, t(F - 2), num), num = round(t(F - 2)) / speed, with 0 put in front and
// F - 1 behind), and the frames are the spline evaluated at those times (motion_spline.py:71-86).
//
//   mg_timewarp_kernel        one wave per candidate: increments exp(mean_t + phi . gamma) in parallel, their running sum by one lane
//                             in canonical-frame order (np's cumulative order: the same bits as mg_time_function_kernel), the
//                             not-a-knot cubic through (t(t'), t') by its second derivatives (a tridiagonal solve, one lane),
//                             the samples in parallel.  The not-a-knot cubic is unique, so this is FITPACK's spline up to rounding
//                             (1e-12 of F on the fixtures); the sample count is int(round(t(F-2)) * (1 / speed)), the value NumPy
//                             before 1.18 made of the float the reference passes (newer NumPy raises TypeError there: SURVEY 8c).
//   mg_frames_at_kernel       one workgroup per candidate: its control points (mean' + E' . s, fma chain over k ascending from the
//                             mean: the arithmetic of mg_back_project_frames_f64) in LDS, the basis row of each of ITS time samples
//                             by the FITPACK recurrence, the 4 taps -- the frames of a candidate at its own times, what
//                             mg_back_project_frames_f64 gives with a time grid per candidate, without a grid per candidate.
#include "mg_internal.h"

#define MG_TW_BLOCK 64
#define MG_TW_MAX_F 2048   // canonical frames the inversion holds in LDS (5 arrays of doubles)

__global__ __launch_bounds__(MG_TW_BLOCK) void mg_timewarp_kernel(const double *__restrict__ tphi, const double *__restrict__ tmean,
                                                                  const void *__restrict__ gamma, int gamma_f64, int64_t ld, int F, int Lt,
                                                                  double inv_speed, double *__restrict__ times, int32_t *__restrict__ lens,
                                                                  int32_t t_cap, double *__restrict__ canonical_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *x = (double *)smem;      // [F] the canonical time function t(t'): the spline's abscissae; ordinates are 0 .. F-1
    double *M = x + F;               // [F] second derivatives
    double *cp = M + F;              // [F] Thomas: modified upper diagonal
    double *dp = cp + F;             // [F] Thomas: modified right-hand side
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const void *gb = gamma_f64 ? (const void *)((const double *)gamma + b * ld) : (const void *)((const float *)gamma + b * ld);
    for (int i = tid; i < F; i += MG_TW_BLOCK) {
        double e = tmean[i];
        for (int l = 0; l < Lt; l++) {
            const double gl = gamma_f64 ? ((const double *)gb)[l] : (double)((const float *)gb)[l];
            e = fma(tphi[(size_t)i * Lt + l], gl, e);
        }
        M[i] = exp(e);               // the increment (parked in M)
    }
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;
        for (int i = 0; i < F; i++) { acc += M[i]; x[i] = acc - 1.0; }
        // not-a-knot cubic through (x_i, i): h_{i-1} M_{i-1} + 2 (h_{i-1} + h_i) M_i + h_i M_{i+1} = 6 ((y_{i+1} - y_i) / h_i - (y_i - y_{i-1}) / h_{i-1}),
        // i = 1 .. F-2, with M_0 and M_{F-1} eliminated by the continuity of the third derivative at x_1 and x_{F-2}
        const int m = F;
        auto h = [&](int i) { return x[i + 1] - x[i]; };
        auto rhs = [&](int i) { return 6.0 * (1.0 / h(i) - 1.0 / h(i - 1)); };   // y_{i+1} - y_i = 1
        // rows 1 .. m-2: (lo, di, up); the first and the last one carry the end conditions
        auto row = [&](int i, double &lo, double &di, double &up) {
            const double h0 = h(i - 1), h1 = h(i);
            if (i == 1) { lo = 0.0; di = 2.0 * (h0 + h1) + h0 * (1.0 + h0 / h1); up = h1 - h0 * h0 / h1; }
            else if (i == m - 2) { lo = h0 - h1 * h1 / h0; di = 2.0 * (h0 + h1) + h1 * (1.0 + h1 / h0); up = 0.0; }
            else { lo = h0; di = 2.0 * (h0 + h1); up = h1; }
        };
        if (m == 4) {   // rows 1 and 2, both with an end condition: row 1 keeps its `up`, row 2 its `lo`
            const double h0 = h(0), h1 = h(1), h2 = h(2);
            const double d1 = 2.0 * (h0 + h1) + h0 * (1.0 + h0 / h1), u1 = h1 - h0 * h0 / h1;
            const double l2 = h1 - h2 * h2 / h1, d2 = 2.0 * (h1 + h2) + h2 * (1.0 + h2 / h1);
            const double r1 = rhs(1), r2 = rhs(2);
            const double det = d1 * d2 - u1 * l2;
            M[1] = (r1 * d2 - u1 * r2) / det;
            M[2] = (d1 * r2 - l2 * r1) / det;
        } else {
            double lo, di, up;
            row(1, lo, di, up);
            cp[1] = up / di;
            dp[1] = rhs(1) / di;
            for (int i = 2; i <= m - 2; i++) {
                row(i, lo, di, up);
                const double den = di - lo * cp[i - 1];
                cp[i] = up / den;
                dp[i] = (rhs(i) - lo * dp[i - 1]) / den;
            }
            M[m - 2] = dp[m - 2];
            for (int i = m - 3; i >= 1; i--) M[i] = dp[i] - cp[i] * M[i + 1];
        }
        {
            const double h0 = h(0), h1 = h(1), a = h(m - 2) / h(m - 3);
            M[0] = (1.0 + h0 / h1) * M[1] - (h0 / h1) * M[2];
            M[m - 1] = (1.0 + a) * M[m - 2] - a * M[m - 3];
        }
    }
    __syncthreads();
    if (canonical_out)
        for (int i = tid; i < F; i += MG_TW_BLOCK) canonical_out[b * F + i] = x[i];
    // the samples: 0, the inverse spline at linspace(1, x[F-2], num), F - 1
    const double stop = x[F - 2];
    const double numf = rint(stop) * inv_speed;
    // a time function that is not finite (exp() overflowed on a wild gamma) or asks for more than 2^24 samples has no row: length 0,
    // nothing written (the conversion of such a float to int is undefined, and T = num + 2 could wrap -- ADVICE r4)
    if (!(numf == numf) || !(fabs(stop) <= 1.0e300) || numf > 16777216.0) {
        if (tid == 0) lens[b] = 0;
        return;
    }
    const int num = numf > 0.0 ? (int)numf : 0;
    const int T = num + 2;
    if (tid == 0) lens[b] = T <= t_cap ? T : -T;   // (negative: the caller's rows are too short; nothing else is written)
    if (T > t_cap) return;
    double *tb = times + b * (int64_t)t_cap;
    const double step = num > 1 ? (stop - 1.0) / (double)(num - 1) : 0.0;
    for (int j = tid; j < T; j += MG_TW_BLOCK) {
        double v;
        if (j == 0) v = 0.0;
        else if (j == T - 1) v = (double)(F - 1);
        else {
            const int q = j - 1;
            const double t = (q == num - 1 && num > 1) ? stop : (double)q * step + 1.0;   // numpy.linspace: arange * step + start, the last one = stop
            int lo = 0, hi = F - 2;                    // the interval [x_i, x_{i+1}) that holds t, the end intervals for anything outside
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (x[mid] <= t) lo = mid; else hi = mid - 1;
            }
            const int i = lo;
            const double hi_ = x[i + 1] - x[i], A = x[i + 1] - t, Bq = t - x[i];
            v = M[i] * A * A * A / (6.0 * hi_) + M[i + 1] * Bq * Bq * Bq / (6.0 * hi_) + ((double)i / hi_ - M[i] * hi_ / 6.0) * A +
                ((double)(i + 1) / hi_ - M[i + 1] * hi_ / 6.0) * Bq;
        }
        tb[j] = v;
    }
}

// FITPACK fpbspl at x on the span l found like splev does (ext = 0): the host's mg_basis_row, statement for statement
__device__ __forceinline__ void mg_basis_row_dev(const double *t, int n, double x, int *i0, double *h) {
    const int k = 3;
    int l = k;
    while (!(x < t[l + 1] || l == n - k - 2)) l++;
    double hh[4];
    h[0] = 1.0; h[1] = h[2] = h[3] = 0.0;
    for (int j = 1; j <= k; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 1; i <= j; i++) {
            const int li = l + i, lj = li - j;
            if (t[li] == t[lj]) { h[i] = 0.0; continue; }
            const double f = hh[i - 1] / (t[li] - t[lj]);
            h[i - 1] = h[i - 1] + f * (t[li] - x);
            h[i] = f * (x - t[lj]);
        }
    }
    *i0 = l - k;
}

#define MG_FA_BLOCK 256
template <bool LAT_F64, bool OUT_F64>
__global__ __launch_bounds__(MG_FA_BLOCK) void mg_frames_at_kernel(const double *__restrict__ Et64,   // [L][R]
                                                                   const double *__restrict__ mean,   // [R]
                                                                   const double *__restrict__ knots,  // [NB + 4]
                                                                   const void *__restrict__ lat, int64_t ld, int L, int R, int D, int NB,
                                                                   const double *__restrict__ times, const int32_t *__restrict__ lens, int32_t t_cap,
                                                                   void *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *c = (double *)smem;                 // [R] the candidate's control points
    double *s = c + R;                          // [L]
    double *w = s + L;                          // [t_cap][4]
    int *i0 = (int *)(w + (size_t)t_cap * 4);   // [t_cap]
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const int T = lens ? lens[b] : t_cap;
    if (T <= 0 || T > t_cap) return;     // (a length beyond the rows' capacity would overrun w / i0 and the row in `out`: the row is skipped)
    for (int k = tid; k < L; k += MG_FA_BLOCK) s[k] = LAT_F64 ? ((const double *)lat)[b * ld + k] : (double)((const float *)lat)[b * ld + k];
    __syncthreads();
    for (int r = tid; r < R; r += MG_FA_BLOCK) {
        double acc = mean[r];
        for (int k = 0; k < L; k++) acc = fma(Et64[(size_t)k * R + r], s[k], acc);
        c[r] = acc;
    }
    const double *tb = times + b * (int64_t)t_cap;
    for (int f = tid; f < T; f += MG_FA_BLOCK) mg_basis_row_dev(knots, NB + 4, tb[f], &i0[f], &w[4 * f]);
    __syncthreads();
    const int64_t TD = (int64_t)T * D;
    for (int64_t e = tid; e < TD; e += MG_FA_BLOCK) {
        const int f = (int)(e / D), d = (int)(e - (int64_t)f * D);
        const double *cf = c + (size_t)i0[f] * D + d;
        const double *wf = w + 4 * f;
        double v = wf[0] * cf[0];
        v = fma(wf[1], cf[D], v);
        v = fma(wf[2], cf[2 * D], v);
        v = fma(wf[3], cf[3 * D], v);
        if (OUT_F64) ((double *)out)[(b * t_cap + f) * D + d] = v;
        else ((float *)out)[(b * t_cap + f) * D + d] = (float)v;
    }
}

int mg_launch_timewarp(mg_primitive *p, const void *gamma, int gdt, int64_t B, int64_t ld, double speed, double *times, int32_t *lens, int32_t t_cap,
                       double *canonical_out) {
    const size_t lds = (size_t)4 * p->F * 8;
    if (p->F < 4 || p->F > MG_TW_MAX_F) { mg_set_error("mg_time_function_sample: %d canonical frames (4 .. %d supported)", p->F, MG_TW_MAX_F); return MG_ERR_UNSUPPORTED; }
    if (!(p->ctx->attr_traj & 2u)) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_timewarp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_at_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_at_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_at_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_at_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        p->ctx->attr_traj |= 2u;
    }
    hipLaunchKernelGGL(mg_timewarp_kernel, dim3((unsigned)B), dim3(MG_TW_BLOCK), lds, p->ctx->stream, (const double *)p->d_tphi, (const double *)p->d_tmean, gamma,
                       gdt == MG_F64 ? 1 : 0, ld, (int)p->F, (int)p->Lt, 1.0 / speed, times, lens, t_cap, canonical_out);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

int mg_launch_frames_at(mg_primitive *p, const void *lat, int ldt, int64_t B, int64_t ld, const double *times, const int32_t *lens, int32_t t_cap,
                        void *out, int odt) {
    const size_t lds = ((size_t)p->R + p->L + (size_t)t_cap * 4) * 8 + (size_t)t_cap * 4 + 16;
    if (lds > 160 * 1024 - 64) {
        mg_set_error("mg_back_project_frames_at: %d control-point rows + %d time samples per candidate do not fit LDS", p->R, t_cap);
        return MG_ERR_UNSUPPORTED;
    }
    if (!(p->ctx->attr_traj & 2u)) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_timewarp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_at_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_at_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_at_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_at_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        p->ctx->attr_traj |= 2u;
    }
    const bool lf = ldt == MG_F64, of = odt == MG_F64;
    hipStream_t st = p->ctx->stream;
#define MG_FA_ARGS (const double *)p->d_Et64, (const double *)p->d_mean, (const double *)p->d_knots, lat, ld, (int)p->L, (int)p->R, (int)p->D, (int)p->NB, times, lens, t_cap, out
    if (lf && of) hipLaunchKernelGGL((mg_frames_at_kernel<true, true>), dim3((unsigned)B), dim3(MG_FA_BLOCK), lds, st, MG_FA_ARGS);
    else if (lf) hipLaunchKernelGGL((mg_frames_at_kernel<true, false>), dim3((unsigned)B), dim3(MG_FA_BLOCK), lds, st, MG_FA_ARGS);
    else if (of) hipLaunchKernelGGL((mg_frames_at_kernel<false, true>), dim3((unsigned)B), dim3(MG_FA_BLOCK), lds, st, MG_FA_ARGS);
    else hipLaunchKernelGGL((mg_frames_at_kernel<false, false>), dim3((unsigned)B), dim3(MG_FA_BLOCK), lds, st, MG_FA_ARGS);
#undef MG_FA_ARGS
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

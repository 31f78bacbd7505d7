// Gaussian-mixture kernels for gfx950 (MI355X), float64 arithmetic.
//
// log p(x) = logsumexp_k [ log w_k + sum log diag P_k - 0.5 L log 2pi - 0.5 |x P_k - mu_k P_k|^2 ]
// which is sklearn's GaussianMixture.score_samples as the reference builds it
// (reference morphablegraphs/motion_model/motion_primitive.py:126-144; formula twin
//  morphablegraphs/motion_model/extended_mgrd_mixture_model.py:60-108).
#include <cstring>

#include <algorithm>

#include "mg_internal.h"
#include "mg_gmm_device.h"
#include "mg_score_device.h"

#define MG_GMM_CANDS 64    // candidates per workgroup: one per lane
#define MG_GMM_WAVES 8     // waves per workgroup: components are dealt round-robin to waves

struct mg_gmm_args {
    const double *P;       // [K][j][i], column j of the upper-triangular P_k contiguous over i <= j
    const double *mP;      // [K][L]   mu_k . P_k
    const double *cst;     // [K]      log w_k + log det - 0.5 L log 2pi
    const void *x;         // (B, ld)
    void *out;             // (B)
    int64_t B, ld;
    int32_t K, L;
};

// The latent tile is staged in LDS ([64][L+1] float64, conflict-free for lane-per-row
// ds_read_b64); every lane owns one candidate, every wave one component at a time, so the
// precision-Cholesky entries are wave-uniform (scalar loads) and the Mahalanobis sum needs
// no cross-lane reduction.  Component terms meet in LDS and lane-owners finish the
// log-sum-exp in component order (the order sklearn/scipy sum in).
template <bool X_F64, bool OUT_F64>
__global__ __launch_bounds__(MG_GMM_CANDS *MG_GMM_WAVES) void mg_gmm_logp_kernel(mg_gmm_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = a.L, K = a.K;
    const int xs = L + 1;
    double *lds_x = (double *)smem;                       // [64][L+1]
    double *lds_t = lds_x + (size_t)MG_GMM_CANDS * xs;    // [K][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * MG_GMM_CANDS;
    const int ncand = (int)((a.B - b0) < MG_GMM_CANDS ? (a.B - b0) : MG_GMM_CANDS);

    for (int e = tid; e < MG_GMM_CANDS * L; e += MG_GMM_CANDS * MG_GMM_WAVES) {
        int c = e / L, i = e - c * L;
        double v = 0.0;
        if (c < ncand) v = X_F64 ? ((const double *)a.x)[(b0 + c) * a.ld + i] : (double)((const float *)a.x)[(b0 + c) * a.ld + i];
        lds_x[c * xs + i] = v;
    }
    __syncthreads();

    const double *xr = lds_x + lane * xs;
    for (int k = wave; k < K; k += MG_GMM_WAVES) {
        const double *Pk = a.P + (size_t)k * L * L;
        const double *mPk = a.mP + (size_t)k * L;
        double maha = 0.0;
        for (int j = 0; j < L; j++) {
            const double *col = Pk + (size_t)j * L;
            double y = -mPk[j];
            for (int i = 0; i <= j; i++) y = fma(xr[i], col[i], y);
            maha = fma(y, y, maha);
        }
        lds_t[k * MG_GMM_CANDS + lane] = a.cst[k] - 0.5 * maha;
    }
    __syncthreads();

    if (wave == 0 && lane < ncand) {
        double vmax = -INFINITY;
        for (int k = 0; k < K; k++) vmax = fmax(vmax, lds_t[k * MG_GMM_CANDS + lane]);
        double r;
        if (vmax == -INFINITY) {
            r = -INFINITY;
        } else {
            double acc = 0.0;
            for (int k = 0; k < K; k++) acc += exp(lds_t[k * MG_GMM_CANDS + lane] - vmax);
            r = log(acc) + vmax;
        }
        if (OUT_F64) ((double *)a.out)[b0 + lane] = r;
        else ((float *)a.out)[b0 + lane] = (float)r;
    }
}

// -----------------------------------------------------------------------------------------
// log_likelihood_jac (reference morphablegraphs/motion_generator/optimization/objective_functions.py:95-107):
//   jac(s) = sum_k N(s | mu_k, Sigma_k) w_k Sigma_k^-1 (s - mu_k) / p(s)          (= -grad log p(s))
// evaluated as sum_k r_k P_k y_k with y_k = (s - mu_k) P_k and r_k = exp(term_k - log p(s)), which is the same
// ratio without the reference's separate underflow of numerator and denominator; where the reference's
// denominator exp(score(s)) underflows to 0 it returns ones, and so does this kernel.
// One workgroup = 16 candidates; y for all components in LDS; float64 throughout.
// -----------------------------------------------------------------------------------------
#define MG_JAC_CANDS 16
#define MG_JAC_ITEMS 8   // (candidate, dimension) outputs per thread: 16 L / 256 rounded up, L <= 128
template <bool X_F64>
__global__ __launch_bounds__(256) void mg_gmm_jac_kernel(mg_gmm_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = a.L, K = a.K, ps = L + 1;
    double *lds_x = (double *)smem;                             // [16][L]
    double *lds_y = lds_x + MG_JAC_CANDS * L;                   // [K][16][L]
    double *lds_t = lds_y + (size_t)K * MG_JAC_CANDS * L;       // [K][16] terms, then responsibilities
    double *lds_lp = lds_t + K * MG_JAC_CANDS;                  // [16] log p
    double *lds_P = lds_lp + MG_JAC_CANDS;                      // [L][L+1]: column j of P_k at j*(L+1), staged per component
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * MG_JAC_CANDS;
    const int ncand = (int)((a.B - b0) < MG_JAC_CANDS ? (a.B - b0) : MG_JAC_CANDS);
    for (int e = tid; e < MG_JAC_CANDS * L; e += 256) {
        const int c = e / L, i = e - c * L;
        double v = 0.0;
        if (c < ncand) v = X_F64 ? ((const double *)a.x)[(b0 + c) * a.ld + i] : (double)((const float *)a.x)[(b0 + c) * a.ld + i];
        lds_x[e] = v;
    }
    auto stage_P = [&](int k) {   // coalesced copy of P_k (column j contiguous over i) into the padded LDS image
        const double *Pk = a.P + (size_t)k * L * L;
        for (int e = tid; e < L * L; e += 256) {
            const int jj = e / L, ii = e - jj * L;
            lds_P[jj * ps + ii] = Pk[e];
        }
    };
    // pass 1: y[k][c][j] = sum_{i <= j} x[c][i] P_k[i][j] - (mu_k P_k)[j]   (P_k upper triangular)
    for (int k = 0; k < K; k++) {
        __syncthreads();
        stage_P(k);
        __syncthreads();
        for (int e = tid; e < MG_JAC_CANDS * L; e += 256) {
            const int c = e / L, jj = e - c * L;
            const double *col = lds_P + jj * ps;
            const double *xr = lds_x + c * L;
            double y = -a.mP[(size_t)k * L + jj];
            for (int i = 0; i <= jj; i++) y = fma(xr[i], col[i], y);
            lds_y[((size_t)k * MG_JAC_CANDS + c) * L + jj] = y;
        }
    }
    __syncthreads();
    for (int e = tid; e < K * MG_JAC_CANDS; e += 256) {
        const double *yr = lds_y + (size_t)e * L;
        double maha = 0.0;
        for (int jj = 0; jj < L; jj++) maha = fma(yr[jj], yr[jj], maha);
        lds_t[e] = a.cst[e / MG_JAC_CANDS] - 0.5 * maha;
    }
    __syncthreads();
    if (tid < MG_JAC_CANDS) {
        double vmax = -INFINITY;
        for (int k = 0; k < K; k++) vmax = fmax(vmax, lds_t[k * MG_JAC_CANDS + tid]);
        double lp = -INFINITY;
        if (vmax != -INFINITY) {
            double acc = 0.0;
            for (int k = 0; k < K; k++) acc += exp(lds_t[k * MG_JAC_CANDS + tid] - vmax);
            lp = log(acc) + vmax;
        }
        lds_lp[tid] = lp;
    }
    __syncthreads();
    for (int e = tid; e < K * MG_JAC_CANDS; e += 256) {
        const double lp = lds_lp[e % MG_JAC_CANDS];
        lds_t[e] = (lp == -INFINITY) ? 0.0 : exp(lds_t[e] - lp);   // responsibility r_k
    }
    // pass 2: jac[c][i] = sum_k r[k][c] sum_{j >= i} P_k[i][j] y[k][c][j], components in order
    double g[MG_JAC_ITEMS];
#pragma unroll
    for (int q = 0; q < MG_JAC_ITEMS; q++) g[q] = 0.0;
    for (int k = 0; k < K; k++) {
        __syncthreads();
        stage_P(k);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MG_JAC_ITEMS; q++) {
            const int e = tid + 256 * q;
            if (e < MG_JAC_CANDS * L) {
                const int c = e / L, i = e - c * L;
                const double *yr = lds_y + ((size_t)k * MG_JAC_CANDS + c) * L;
                double z = 0.0;
                for (int jj = i; jj < L; jj++) z = fma(lds_P[jj * ps + i], yr[jj], z);
                g[q] = fma(lds_t[k * MG_JAC_CANDS + c], z, g[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MG_JAC_ITEMS; q++) {
        const int e = tid + 256 * q;
        if (e < MG_JAC_CANDS * L) {
            const int c = e / L, i = e - c * L;
            if (c < ncand)   // the reference: denominator exp(score) == 0 -> np.ones(s.shape)
                ((double *)a.out)[(b0 + c) * L + i] = (exp(lds_lp[c]) == 0.0) ? 1.0 : g[q];
        }
    }
}


// -----------------------------------------------------------------------------------------
// MFMA variant (n_components <= 64): one workgroup = 16 candidates, one wave = one mixture
// component at a time.  Y = X P_k - mu_k P_k by v_mfma_f64_16x16x4_f64: A = the latent tile
// (registers), B = precision-Cholesky fragments streamed from L2 (only the k-steps at or above
// the diagonal of each 16-column tile), C-in = -mu_k P_k.  Each lane squares its 4 results,
// sums over the column tiles, and the Mahalanobis term is finished by a wavefront
// (butterfly) reduction over the 16 lanes that share the candidate rows.
// -----------------------------------------------------------------------------------------

struct mg_gmm_mfma_args {
    int64_t B, ld;
    int32_t K, L, JT;
};

template <int KK, bool X_F64, bool OUT_F64>
__global__ __launch_bounds__(256) void mg_gmm_logp_mfma_kernel(const double *__restrict__ Ppack,  // [K][JT][KK][64]
                                                              const double *__restrict__ mP,     // [K][JT*16]
                                                              const double *__restrict__ cst,    // [K]
                                                              const void *__restrict__ x, void *__restrict__ out,
                                                              const mg_gmm_mfma_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mg_lds_f64 *lds_t = (mg_lds_f64 *)smem;   // [K][16] component terms
    mg_lds_f64 *lds_e = lds_t + a.K * 16;     // [K][16] exp(term - max)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;
    const int64_t b0 = (int64_t)blockIdx.x * 16;
    const int ncand = (int)((a.B - b0) < 16 ? (a.B - b0) : 16);
    typename mg_gmm_xt<X_F64>::type xf[KK];
    mg_gmm_load_x<KK, X_F64>(xf, x, b0, ncand, a.ld, a.L, cl, g);
    for (int k = wave; k < a.K; k += 4) {
        mg_gmm_frag<KK> f;
        mg_gmm_load_component<KK>(f, Ppack, mP, cst, k, a.JT, lane, cl, a.K);
        mg_gmm_apply_component(f, k, a.JT, xf, lds_t, cl, g);
    }
    __syncthreads();
    // log-sum-exp: the K exponentials of a candidate in parallel, summed in component order
    for (int e = tid; e < a.K * 16; e += 256) lds_e[e] = mg_gmm_exp_entry(lds_t, a.K, e);
    __syncthreads();
    if (tid < ncand) {
        const double r = mg_gmm_logsumexp(lds_t, lds_e, a.K, tid);
        if (OUT_F64) ((double *)out)[b0 + tid] = r;
        else ((float *)out)[b0 + tid] = (float)r;
    }
}

template <int KK>
static int mg_launch_gmm_mfma_kk(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, void *out, int odt) {
    mg_gmm_mfma_args a;
    a.B = B; a.ld = ld; a.K = p->K; a.L = p->Lg; a.JT = (p->Lg + 15) / 16;
    const int64_t grid = (B + 15) / 16;
    if (grid > 0x7fffffff) { mg_set_error("mg_gmm_log_prob: too many samples"); return MG_ERR_UNSUPPORTED; }
    const size_t lds = (size_t)p->K * 16 * 8 * 2;
    hipStream_t st = p->ctx->stream;
    const bool xf = xdt == MG_F64, of = odt == MG_F64;
    if (xf && of) hipLaunchKernelGGL((mg_gmm_logp_mfma_kernel<KK, true, true>), dim3((int)grid), dim3(256), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a);
    else if (xf) hipLaunchKernelGGL((mg_gmm_logp_mfma_kernel<KK, true, false>), dim3((int)grid), dim3(256), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a);
    else if (of) hipLaunchKernelGGL((mg_gmm_logp_mfma_kernel<KK, false, true>), dim3((int)grid), dim3(256), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a);
    else hipLaunchKernelGGL((mg_gmm_logp_mfma_kernel<KK, false, false>), dim3((int)grid), dim3(256), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// -----------------------------------------------------------------------------------------
// Large batches (the optimizer's 131 072 candidates per iteration, BASELINE configs[4]): the kernel above re-reads the
// mixture's fragments from L2 for every 16 candidates (102 KB per tile: 838 MB per launch at B = 131 072) and its matrix
// pipe waits for them (0.48 of the float64 matrix peak).  Here a persistent 1024-thread workgroup per CU stages the
// fragments the chains use (the k-steps at or above each column tile's diagonal: 22 of 30 for L = 40) in LDS ONCE -- 90 KB
// for K = 8 -- and its sixteen waves walk 16-candidate tiles, every wave a tile of its own through all K components:
// B operand = one conflict-free ds_read_b64 per MFMA, four waves per SIMD cover each other's LDS latency.  The chains,
// the squares, the butterfly, the log-sum-exp are the functions the kernel above uses, in the same order: same bits.
// -----------------------------------------------------------------------------------------
template <int KK>
__host__ __device__ constexpr int mg_gmm_nf() {
    int nf = 0;
    for (int jt = 0; jt < (KK + 3) / 4; jt++) nf += (4 * (jt + 1) < KK) ? 4 * (jt + 1) : KK;
    return nf;
}
// SCORE (mg_objective_error_and_naturalness): the optimiser's objective in ONE launch.  The wave that holds a tile's latents for
// the mixture also scores the keyframe constraints on them -- the pose channels X . W^T + bias on the f64 matrix pipe, residuals by
// mg_constraint_residual, summed in constraint order: the arithmetic of mg_score_mfma_kernel, so the same bits -- and writes
// error_scale * error + quality_scale * (-log p) beside (or instead of) the two parts; the 21 MB of latents are read once.
struct mg_objective_args {
    mg_score_args sa;            // sa.out unused
    const double *Wpack, *bpad;  // [RT][KK][64], [RT*16]
    int32_t RT, rows, wave_doubles;   // wave_doubles: the wave's LDS buffer (terms + exponentials, then reused for channels + residuals)
    int32_t pair_doubles;             // > 0: a second channel buffer per wave behind the sixteen wave buffers -- the residuals of TWO tiles in one pass
    double error_scale, quality_scale;
    double *err_out, *obj_out;   // (B) float64 each, or NULL
};
// Waves per workgroup: sixteen for the mixture alone (four per SIMD cover each other's fragment reads), TWELVE with SCORE: at four
// waves per SIMD a wave has 128 registers, and the scoring's state on top of the mixture's (120) spilled 19 of them into the
// store stream's way; at three it has 170, nothing spills, and the matrix pipe is as busy (the mixture alone runs in the same
// 65.5 us with twelve waves; with eight it loses 5 us).  131 072 candidates: 75.5 -> 73.2 us.
template <bool SCORE> struct mg_gmm_lds_nw { static constexpr int value = SCORE ? 12 : 16; };
template <int KK, bool X_F64, bool OUT_F64, bool SCORE>
__global__ __launch_bounds__(64 * mg_gmm_lds_nw<SCORE>::value) void mg_gmm_logp_lds_kernel(const double *__restrict__ Ppack,  // [K][JT][KK][64]
                                                               const double *__restrict__ mP,     // [K][JT*16]
                                                               const double *__restrict__ cst,    // [K]
                                                               const void *__restrict__ x, void *__restrict__ out,
                                                               const mg_gmm_mfma_args a, const int64_t n_tiles, const mg_objective_args oa) {
    constexpr int JTM = (KK + 3) / 4;
    constexpr int NW = mg_gmm_lds_nw<SCORE>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K;
    constexpr int JT = JTM;                         // KK is the smallest even number of k-steps for L: ceil(L / 16) == JTM (checked at the launch)
    constexpr int nf = mg_gmm_nf<KK>();             // fragments per component that the chains use
    mg_lds_f64 *lds_f = (mg_lds_f64 *)smem;         // [K][nf][64]: component k, column tile jt, k-step kk at ((k nf + off(jt) + kk) 64 + lane)
    mg_lds_f64 *lds_c = lds_f + (size_t)K * nf * 64;   // [K][JT*16]: -mu_k P_k
    mg_lds_f64 *lds_w = lds_c + (size_t)K * JT * 16;   // per wave: [K][16] terms, [K][16] exponentials
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;
    // staging: a wave copies whole 512-byte fragments (wave-uniform index arithmetic, several loads in flight)
    // (from the paired copy behind the image, mg_host.hip: two k-steps per 16-byte load; a column tile's count of k-steps is even, so is nf)
    typedef double mg_f64x2 __attribute__((ext_vector_type(2)));
    const mg_f64x2 *P2 = (const mg_f64x2 *)(Ppack + (size_t)K * JT * KK * 64);
#pragma unroll 4
    for (int F2 = wave; F2 < K * nf / 2; F2 += NW) {
        const int F = 2 * F2, k = F / nf, f = F - k * nf;
        int jt = 0, off = 0;
#pragma unroll
        for (int j = 0; j < JTM - 1; j++) {
            const int n = (4 * (j + 1) < KK) ? 4 * (j + 1) : KK;
            if (f >= off + n) { off += n; jt = j + 1; }
        }
        const mg_f64x2 v = P2[(((size_t)k * JT + jt) * (KK / 2) + (f - off) / 2) * 64 + lane];
        lds_f[F * 64 + lane] = v[0];
        lds_f[(F + 1) * 64 + lane] = v[1];
    }
    for (int e = tid; e < K * JT * 16; e += 64 * NW) lds_c[e] = -mP[e];
    __syncthreads();
    mg_lds_f64 *terms = lds_w + (size_t)wave * (SCORE ? oa.wave_doubles : 2 * K * 16), *exps = terms + K * 16;
    // SCORE, pairs: the residual lanes are the kernel's vector work (a direction residual is five roots, six divisions and an arc cosine
    // in float64; vector work on a SIMD takes the matrix pipe's issue slots) and a tile fills 16 n of a wave's 64 lanes.  A wave therefore
    // parks the channels of every other tile in a second buffer and scores two tiles' candidates in one pass: half the instructions.
    double *vals_b = SCORE && oa.pair_doubles > 0 ? (double *)(lds_w + (size_t)NW * oa.wave_doubles + (size_t)wave * oa.pair_doubles) : nullptr;
    bool pending = false;
    double r_a = 0.0;
    int64_t b0_a = 0;
    int ncand_a = 0;
    // tile t of the launch: workgroup t % grid, wave (t / grid) % 16 -- consecutive tiles go to different CUs.  (Requesting the
    // next tile's latents a tile ahead changes nothing: the other three waves of the SIMD cover the load.)
    for (int64_t tile = (int64_t)wave * gridDim.x + blockIdx.x; tile < n_tiles; tile += (int64_t)gridDim.x * NW) {
        const int64_t b0 = tile * 16;
        const int ncand = (int)((a.B - b0) < 16 ? (a.B - b0) : 16);
        typename mg_gmm_xt<X_F64>::type xf[KK];
        mg_gmm_load_x<KK, X_F64>(xf, x, b0, ncand, a.ld, a.L, cl, g);
        for (int k = 0; k < K; k++) {
            __builtin_amdgcn_sched_barrier(0);   // a component's fragment reads stay inside its iteration: the waves of a SIMD cover each other
            const mg_lds_f64 *fk = lds_f + (size_t)k * nf * 64 + lane;
            mg_f64x4 acc[JTM];
#pragma unroll
            for (int jt = 0; jt < JTM; jt++) {
                const double c0 = lds_c[(k * JT + jt) * 16 + cl];
                acc[jt] = {c0, c0, c0, c0};
            }
#pragma unroll
            for (int kk = 0; kk < KK; kk++) {
                int off = 0;
#pragma unroll
                for (int jt = 0; jt < JTM; jt++) {
                    if (kk < 4 * (jt + 1)) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xf[kk], fk[(off + kk) * 64], acc[jt], 0, 0, 0);
                    off += (4 * (jt + 1) < KK) ? 4 * (jt + 1) : KK;
                }
            }
            mg_gmm_finish_component<JTM>(acc, JT, cst[k], k, terms, cl, g);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its LDS writes are visible to its reads in order
        for (int e = lane; e < K * 16; e += 64) exps[e] = mg_gmm_exp_entry(terms, K, e);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double r = 0.0;
        if (lane < ncand) {
            r = mg_gmm_logsumexp(terms, exps, K, lane);
            if (out) {
                if (OUT_F64) ((double *)out)[b0 + lane] = r;
                else ((float *)out)[b0 + lane] = (float)r;
            }
        }
        if constexpr (SCORE) {
            // the wave's buffer again, now for the tile's pose channels and residuals (LDS serves a wave's requests in order)
            const int RT = oa.RT, rows = oa.rows, vs = rows + 1, n = oa.sa.n;   // (only the rows in use are kept: sixteen waves share the LDS)
            const bool park = vals_b != nullptr && !pending && tile + (int64_t)gridDim.x * NW < n_tiles;   // another tile follows: score the two together
            double *vals = park ? vals_b : (double *)terms, *resid = (double *)terms + 16 * vs;           // resid: [2][n][16]
            for (int rt = 0; rt < RT; rt++) {
                const double *wp = oa.Wpack + ((size_t)rt * KK) * 64 + lane;
                const double c0 = oa.bpad[rt * 16 + cl];
                mg_f64x4 acc = {c0, c0, c0, c0};
#pragma unroll
                for (int kk = 0; kk < KK; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xf[kk], wp[kk * 64], acc, 0, 0, 0);
                if (rt * 16 + cl < rows) {
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) vals[(g + 4 * rr) * vs + rt * 16 + cl] = acc[rr];
                }
            }
            if (park) {
                pending = true; r_a = r; b0_a = b0; ncand_a = ncand;
                continue;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int per = 16 * n, items = pending ? 2 * per : per;      // the parked tile's candidates first, then this tile's
            for (int e = lane; e < items; e += 64) {
                const int second = (pending && e >= per) ? 1 : 0, rem = e - second * per;
                const int cand = rem & 15, c = rem >> 4;
                const bool from_b = pending && !second;
                const double *v = (from_b ? vals_b : (const double *)terms) + cand * vs;
                resid[(second * n + c) * 16 + cand] = mg_constraint_residual<true>(oa.sa, c, [&](int row) { return v[row]; }, (from_b ? b0_a : b0) + cand);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                const int second = (pending && lane >= 16) ? 1 : 0, cand = lane & 15;
                const bool mine_a = pending && !second;
                const int nc = mine_a ? ncand_a : ncand;
                // (the log-likelihood of candidate `cand` sits in lane cand: r of this tile, r_a of the parked one)
                const double lp_b = __shfl(r, cand), lp_a = __shfl(r_a, cand);
                if (lane < (pending ? 32 : 16) && cand < nc) {
                    const int64_t bb = (mine_a ? b0_a : b0) + cand;
                    const double lp = mine_a ? lp_a : lp_b;
                    double err = 0.0;
                    for (int c = 0; c < n; c++) err += resid[(second * n + c) * 16 + cand];
                    if (oa.err_out) oa.err_out[bb] = err;
                    if (oa.obj_out) {
                        const double e_part = oa.error_scale * err, q_part = -lp * oa.quality_scale;   // (rounded separately, like the host's array arithmetic)
                        oa.obj_out[bb] = e_part + q_part;
                    }
                }
            }
            pending = false;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the next tile's terms land in the same buffer
        }
    }
}

static int mg_gmm_lds_nf(int KK, int JT) {
    int nf = 0;
    for (int jt = 0; jt < JT; jt++) nf += std::min(4 * (jt + 1), KK);
    return nf;
}
// the LDS-resident kernel pays from this many candidates on ('walk', us per launch, one tile per workgroup / LDS-resident:
// B = 8192 9.9 / 15.6, 16384 16.6 / 16.9, 32768 30.2 / 24.8, 65536 51.0 / 38.9, 131072 91.0 / 67.4)
#define MG_GMM_LDS_MIN_B 20480
static int mg_objective_wave_doubles(const mg_primitive *p, const mg_constraint_set *cs) {
    return std::max(2 * p->K * 16, 16 * (cs->rows + 1) + 32 * std::max(cs->n, 1));   // channels [16][rows + 1], residuals [2][n][16]
}
template <int KK>
static int mg_launch_gmm_lds_kk(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, void *out, int odt,
                                const mg_constraint_set *cs = nullptr, double error_scale = 0.0, double quality_scale = 0.0,
                                double *err_out = nullptr, double *obj_out = nullptr) {
    mg_gmm_mfma_args a;
    a.B = B; a.ld = ld; a.K = p->K; a.L = p->Lg; a.JT = (p->Lg + 15) / 16;
    const int64_t n_tiles = (B + 15) / 16;
    mg_objective_args oa;
    memset(&oa, 0, sizeof(oa));
    const int wave_doubles = cs ? mg_objective_wave_doubles(p, cs) : 2 * p->K * 16;
    const int nw = cs ? mg_gmm_lds_nw<true>::value : mg_gmm_lds_nw<false>::value;
    size_t lds = ((size_t)p->K * mg_gmm_lds_nf(KK, a.JT) * 64 + (size_t)p->K * a.JT * 16 + (size_t)nw * wave_doubles) * 8;
    const int pair_doubles = cs ? 16 * (cs->rows + 1) : 0;
    const bool pairs = cs && lds + (size_t)nw * pair_doubles * 8 <= 160 * 1024;   // room for the second channel buffers: two tiles per residual pass
    if (pairs) lds += (size_t)nw * pair_doubles * 8;
    if (cs) {
        mg_score_args &sa = oa.sa;
        sa.W = cs->d_W; sa.bias = cs->d_bias; sa.par = cs->d_par; sa.woff = cs->d_woff; sa.chain = cs->d_chain; sa.choff = cs->d_choff;
        sa.align = cs->d_align; sa.align_cand = nullptr; sa.pose = cs->d_pose; sa.lat = x; sa.out = nullptr; sa.res = nullptr; sa.B = B; sa.ld = ld;
        sa.n = cs->n; sa.nch = cs->nch; sa.L = p->L;
        oa.Wpack = cs->d_Wpack; oa.bpad = cs->d_bpad; oa.RT = cs->RT; oa.rows = cs->rows; oa.wave_doubles = wave_doubles;
        oa.pair_doubles = pairs ? pair_doubles : 0;
        oa.error_scale = error_scale; oa.quality_scale = quality_scale; oa.err_out = err_out; oa.obj_out = obj_out;
    }
    // every CU gets a workgroup as soon as there are that many tiles: the tiles of a workgroup run side by side on its sixteen
    // waves, so few tiles per workgroup mean short chains per SIMD (16 workgroups of 16 busy waves: 36 us at B = 4096)
    const int grid = (int)std::min<int64_t>(std::max(1, p->ctx->n_cu), n_tiles);
    hipStream_t st = p->ctx->stream;
    const bool xf = xdt == MG_F64, of = odt == MG_F64;
    // the dynamic-LDS attribute is a property of (kernel, device): once per context (a per-process flag left a second
    // GPU's context without it)
    const unsigned abit = 1u << (KK / 2);
    if (!(p->ctx->attr_gmm_lds & abit)) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_lds_kernel<KK, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_lds_kernel<KK, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_lds_kernel<KK, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_lds_kernel<KK, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_lds_kernel<KK, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_lds_kernel<KK, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        p->ctx->attr_gmm_lds |= abit;
    }
    if (cs) {   // the objective: log p in float64 (or not at all), errors and objective in float64
        if (xf) hipLaunchKernelGGL((mg_gmm_logp_lds_kernel<KK, true, true, true>), dim3(grid), dim3(64 * nw), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a, n_tiles, oa);
        else hipLaunchKernelGGL((mg_gmm_logp_lds_kernel<KK, false, true, true>), dim3(grid), dim3(64 * nw), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a, n_tiles, oa);
    }
    else if (xf && of) hipLaunchKernelGGL((mg_gmm_logp_lds_kernel<KK, true, true, false>), dim3(grid), dim3(1024), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a, n_tiles, oa);
    else if (xf) hipLaunchKernelGGL((mg_gmm_logp_lds_kernel<KK, true, false, false>), dim3(grid), dim3(1024), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a, n_tiles, oa);
    else if (of) hipLaunchKernelGGL((mg_gmm_logp_lds_kernel<KK, false, true, false>), dim3(grid), dim3(1024), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a, n_tiles, oa);
    else hipLaunchKernelGGL((mg_gmm_logp_lds_kernel<KK, false, false, false>), dim3(grid), dim3(1024), lds, st, p->d_gPpack, p->d_gmPpad, p->d_gconst, x, out, a, n_tiles, oa);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// The optimiser's objective in one launch: can the LDS-resident mixture kernel carry this constraint set?
bool mg_objective_can_fuse(const mg_primitive *p, const mg_constraint_set *cs) {
    if (!p->d_gPpack || p->KKg <= 0 || p->Lg != p->L || p->KK != p->KKg || !cs || !cs->d_Wpack || p->ctx->opt[MG_OPT_FORCE_VALU_SCORE]) return false;
    // root position / 2-D direction constraints (path following: what the optimiser's keyframe constraints are when no hand or
    // foot is constrained), aligned to the root joint if at all: the kinds the kernel compiles in
    for (const mg_keyframe_constraint &kc : cs->structure)
        if (kc.type != MG_CONSTRAINT_POSITION && kc.type != MG_CONSTRAINT_DIRECTION_2D) return false;
    if (cs->has_pose || cs->align_joint > 0) return false;
    const int JT = (p->Lg + 15) / 16;
    if (JT != (p->KKg + 3) / 4) return false;
    const size_t lds = ((size_t)p->K * mg_gmm_lds_nf(p->KKg, JT) * 64 + (size_t)p->K * JT * 16 + (size_t)mg_gmm_lds_nw<true>::value * mg_objective_wave_doubles(p, cs)) * 8;
    return lds <= 160 * 1024 - 64;
}
int mg_launch_objective(mg_primitive *p, const mg_constraint_set *cs, const void *x, int xdt, int64_t B, int64_t ld, double error_scale,
                        double quality_scale, double *logp_out, double *err_out, double *obj_out) {
    switch (p->KKg) {
        case 2: return mg_launch_gmm_lds_kk<2>(p, x, xdt, B, ld, logp_out, MG_F64, cs, error_scale, quality_scale, err_out, obj_out);
        case 4: return mg_launch_gmm_lds_kk<4>(p, x, xdt, B, ld, logp_out, MG_F64, cs, error_scale, quality_scale, err_out, obj_out);
        case 6: return mg_launch_gmm_lds_kk<6>(p, x, xdt, B, ld, logp_out, MG_F64, cs, error_scale, quality_scale, err_out, obj_out);
        case 8: return mg_launch_gmm_lds_kk<8>(p, x, xdt, B, ld, logp_out, MG_F64, cs, error_scale, quality_scale, err_out, obj_out);
        case 10: return mg_launch_gmm_lds_kk<10>(p, x, xdt, B, ld, logp_out, MG_F64, cs, error_scale, quality_scale, err_out, obj_out);
        case 12: return mg_launch_gmm_lds_kk<12>(p, x, xdt, B, ld, logp_out, MG_F64, cs, error_scale, quality_scale, err_out, obj_out);
        case 14: return mg_launch_gmm_lds_kk<14>(p, x, xdt, B, ld, logp_out, MG_F64, cs, error_scale, quality_scale, err_out, obj_out);
        case 16: return mg_launch_gmm_lds_kk<16>(p, x, xdt, B, ld, logp_out, MG_F64, cs, error_scale, quality_scale, err_out, obj_out);
        default: break;
    }
    mg_set_error("mg_objective_error_and_naturalness: no one-launch kernel for this mixture");
    return MG_ERR_UNSUPPORTED;
}
static bool mg_gmm_use_lds_kernel(const mg_primitive *p, int64_t B) {
    const int mode = p->ctx->opt[MG_OPT_GMM_KERNEL];   // 0 = by batch size, 1 = one tile per workgroup, 2 = LDS-resident
    if (mode == 1 || !p->d_gPpack || p->KKg <= 0) return false;
    const int JT = (p->Lg + 15) / 16;
    if (JT != (p->KKg + 3) / 4) return false;       // (cannot happen: KKg is the smallest even number of k-steps for Lg)
    const size_t lds = ((size_t)p->K * mg_gmm_lds_nf(p->KKg, JT) * 64 + (size_t)p->K * JT * 16 + (size_t)16 * 2 * p->K * 16) * 8;
    if (lds > 160 * 1024 - 64) return false;
    return mode == 2 || B >= MG_GMM_LDS_MIN_B;
}

int mg_launch_gmm_logp(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, void *out, int odt) {
    if (mg_gmm_use_lds_kernel(p, B)) {
        switch (p->KKg) {
            case 2: return mg_launch_gmm_lds_kk<2>(p, x, xdt, B, ld, out, odt);
            case 4: return mg_launch_gmm_lds_kk<4>(p, x, xdt, B, ld, out, odt);
            case 6: return mg_launch_gmm_lds_kk<6>(p, x, xdt, B, ld, out, odt);
            case 8: return mg_launch_gmm_lds_kk<8>(p, x, xdt, B, ld, out, odt);
            case 10: return mg_launch_gmm_lds_kk<10>(p, x, xdt, B, ld, out, odt);
            case 12: return mg_launch_gmm_lds_kk<12>(p, x, xdt, B, ld, out, odt);
            case 14: return mg_launch_gmm_lds_kk<14>(p, x, xdt, B, ld, out, odt);
            case 16: return mg_launch_gmm_lds_kk<16>(p, x, xdt, B, ld, out, odt);
            default: break;
        }
    }
    if (p->d_gPpack && p->K * 16 * 16 <= 60 * 1024) {
        switch (p->KKg) {
            case 2: return mg_launch_gmm_mfma_kk<2>(p, x, xdt, B, ld, out, odt);
            case 4: return mg_launch_gmm_mfma_kk<4>(p, x, xdt, B, ld, out, odt);
            case 6: return mg_launch_gmm_mfma_kk<6>(p, x, xdt, B, ld, out, odt);
            case 8: return mg_launch_gmm_mfma_kk<8>(p, x, xdt, B, ld, out, odt);
            case 10: return mg_launch_gmm_mfma_kk<10>(p, x, xdt, B, ld, out, odt);
            case 12: return mg_launch_gmm_mfma_kk<12>(p, x, xdt, B, ld, out, odt);
            case 14: return mg_launch_gmm_mfma_kk<14>(p, x, xdt, B, ld, out, odt);
            case 16: return mg_launch_gmm_mfma_kk<16>(p, x, xdt, B, ld, out, odt);
            default: break;
        }
    }
    mg_gmm_args a;
    a.P = p->d_gP; a.mP = p->d_gmP; a.cst = p->d_gconst; a.x = x; a.out = out; a.B = B; a.ld = ld; a.K = p->K; a.L = p->Lg;
    int64_t grid = (B + MG_GMM_CANDS - 1) / MG_GMM_CANDS;
    size_t lds = ((size_t)MG_GMM_CANDS * (p->Lg + 1) + (size_t)p->K * MG_GMM_CANDS) * 8;
    if (lds > 150 * 1024 || grid > 0x7fffffff) {
        mg_set_error("mg_gmm_log_prob: n_components %d / n_gmm %d too large for the LDS-staged kernel", p->Lg, p->K);
        return MG_ERR_UNSUPPORTED;
    }
    dim3 blk(MG_GMM_CANDS * MG_GMM_WAVES);
    hipStream_t st = p->ctx->stream;
    const bool xf = xdt == MG_F64, of = odt == MG_F64;
    if (lds > 64 * 1024) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_logp_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (xf && of) hipLaunchKernelGGL((mg_gmm_logp_kernel<true, true>), dim3((int)grid), blk, lds, st, a);
    else if (xf) hipLaunchKernelGGL((mg_gmm_logp_kernel<true, false>), dim3((int)grid), blk, lds, st, a);
    else if (of) hipLaunchKernelGGL((mg_gmm_logp_kernel<false, true>), dim3((int)grid), blk, lds, st, a);
    else hipLaunchKernelGGL((mg_gmm_logp_kernel<false, false>), dim3((int)grid), blk, lds, st, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// -----------------------------------------------------------------------------------------
// Sampler: x = mu_c + chol(Sigma_c) z,  z ~ N(0, I) from Philox4x32-10 + Box-Muller.
// Rows [cum[c], cum[c+1]) belong to component c (sklearn groups rows by component,
// reference motion_primitive.py:182-189).  Not bit-compatible with sklearn's stream.
// -----------------------------------------------------------------------------------------
struct mg_sample_args {
    const double *chol;   // [K][L][L] lower
    const double *mean;   // [K][L]
    const int64_t *cum;   // [K+1]
    void *x;
    int32_t *comp;
    int64_t n, ld;
    uint64_t seed;
    int32_t K, L;
    int64_t row_lo, row_hi;   // the rows of the global draw this launch produces: x row (b - row_lo)
};

#define MG_SAMPLE_BLOCK 64   // one wave per workgroup: 8192 samples spread over 128 CUs (the kernel is latency bound)
template <bool X_F64>
__global__ __launch_bounds__(MG_SAMPLE_BLOCK) void mg_gmm_sample_kernel(mg_sample_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = a.L, zs = L + 1;
    double *lds_z = (double *)smem;   // [MG_SAMPLE_BLOCK][L+1]
    const int tid = threadIdx.x;
    const int64_t b = a.row_lo + (int64_t)blockIdx.x * MG_SAMPLE_BLOCK + tid;
    if (b < a.row_hi) {
        double *z = lds_z + (size_t)tid * zs;
        for (int i = 0; i < L; i += 4) {
            double zz[4];
            mg_normal4(b, i >> 2, a.seed, zz);
            for (int q = 0; q < 4 && i + q < L; q++) z[i + q] = zz[q];
        }
        int c = 0;
        while (c + 1 < a.K && b >= a.cum[c + 1]) c++;
        const double *Lc = a.chol + (size_t)c * L * L;
        const double *mu = a.mean + (size_t)c * L;
        for (int i = 0; i < L; i++) {
            double acc = mu[i];
            for (int j = 0; j <= i; j++) acc = fma(Lc[i * L + j], z[j], acc);
            if (X_F64) ((double *)a.x)[(b - a.row_lo) * a.ld + i] = acc;
            else ((float *)a.x)[(b - a.row_lo) * a.ld + i] = (float)acc;
        }
        if (a.comp) a.comp[b - a.row_lo] = c;
    }
}

int mg_launch_gmm_sample_valu(mg_primitive *p, int64_t n, const int64_t *cum_dev, uint64_t seed, void *x, int xdt, int64_t ld, int32_t *comp,
                              int64_t row_lo, int64_t row_hi) {
    mg_sample_args a;
    a.chol = p->d_gchol; a.mean = p->d_gmean; a.cum = cum_dev; a.x = x; a.comp = comp; a.n = n; a.ld = ld; a.seed = seed; a.K = p->K; a.L = p->Lg;
    a.row_lo = row_lo; a.row_hi = row_hi;
    n = row_hi - row_lo;
    size_t lds = (size_t)MG_SAMPLE_BLOCK * (p->Lg + 1) * 8;
    if (lds > 150 * 1024) { mg_set_error("mg_gmm_sample: n_components %d too large", p->Lg); return MG_ERR_UNSUPPORTED; }
    int64_t grid = (n + MG_SAMPLE_BLOCK - 1) / MG_SAMPLE_BLOCK;
    if (grid > 0x7fffffff) { mg_set_error("mg_gmm_sample: too many samples"); return MG_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_sample_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_sample_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (xdt == MG_F64) hipLaunchKernelGGL((mg_gmm_sample_kernel<true>), dim3((int)grid), dim3(MG_SAMPLE_BLOCK), lds, p->ctx->stream, a);
    else hipLaunchKernelGGL((mg_gmm_sample_kernel<false>), dim3((int)grid), dim3(MG_SAMPLE_BLOCK), lds, p->ctx->stream, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// MFMA form of the same Jacobian for n_components <= 64: y_k = x P_k - mu_k P_k as in the log-likelihood kernel
// (kept in LDS this time), responsibilities from the log-sum-exp, then z_k = y_k P_k^T by a second chain of
// v_mfma_f64_16x16x4_f64 against the transposed fragments (P_k^T is lower triangular: column tile `it` needs only
// the k-steps >= 4 it), jac = sum_k r_k z_k.  16 candidates per workgroup, one component per wave at a time; the
// four waves' partial sums meet in LDS and are added in wave order.
template <int KK, bool X_F64>
__global__ __launch_bounds__(256) void mg_gmm_jac_mfma_kernel(const double *__restrict__ Ppack,    // [K][JT][KK][64]
                                                             const double *__restrict__ PTpack,   // [K][JT][KK][64]
                                                             const double *__restrict__ mP,       // [K][JT*16]
                                                             const double *__restrict__ cst,      // [K]
                                                             const void *__restrict__ x, double *__restrict__ out,
                                                             const mg_gmm_mfma_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int JTM = (KK + 3) / 4;
    constexpr int YS = JTM * 16 + 1;                       // padded row of y: [cand][j], j < 16 JTM (zero beyond L)
    mg_lds_f64 *lds_y = (mg_lds_f64 *)smem;               // [K][16][YS]
    mg_lds_f64 *lds_t = lds_y + a.K * 16 * YS;             // [K][16] terms, then responsibilities
    mg_lds_f64 *lds_lp = lds_t + a.K * 16;                 // [16]
    mg_lds_f64 *lds_g = lds_lp + 16;                       // [4 waves][16][JTM*16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;
    const int64_t b0 = (int64_t)blockIdx.x * 16;
    const int ncand = (int)((a.B - b0) < 16 ? (a.B - b0) : 16);
    typename mg_gmm_xt<X_F64>::type xf[KK];
    mg_gmm_load_x<KK, X_F64>(xf, x, b0, ncand, a.ld, a.L, cl, g);
    for (int k = wave; k < a.K; k += 4) {
        mg_gmm_frag<KK> f;
        mg_gmm_load_component<KK>(f, Ppack, mP, cst, k, a.JT, lane, cl, a.K);
        mg_f64x4 acc[JTM];
#pragma unroll
        for (int jt = 0; jt < JTM; jt++) acc[jt] = {f.c0[jt], f.c0[jt], f.c0[jt], f.c0[jt]};
#pragma unroll
        for (int kk = 0; kk < KK; kk++)
#pragma unroll
            for (int jt = 0; jt < JTM; jt++)
                if (kk < 4 * (jt + 1)) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xf[kk], f.pf[jt][kk], acc[jt], 0, 0, 0);
        double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int jt = 0; jt < JTM; jt++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                // C layout: col j = 16 jt + cl, row (candidate) = g + 4 r; columns >= L are zero (zero fragments, zero C-in)
                lds_y[(k * 16 + g + 4 * r) * YS + 16 * jt + cl] = acc[jt][r];
                if (jt < a.JT) part[r] = fma(acc[jt][r], acc[jt][r], part[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = part[r];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            part[r] = v;
        }
        if (cl == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) lds_t[k * 16 + g + 4 * r] = f.cst - 0.5 * part[r];
        }
    }
    __syncthreads();
    if (tid < 16) {
        double vmax = -INFINITY;
        for (int k = 0; k < a.K; k++) vmax = fmax(vmax, lds_t[k * 16 + tid]);
        double lp = -INFINITY;
        if (vmax != -INFINITY) {
            double s = 0.0;
            for (int k = 0; k < a.K; k++) s += exp(lds_t[k * 16 + tid] - vmax);
            lp = log(s) + vmax;
        }
        lds_lp[tid] = lp;
    }
    __syncthreads();
    for (int e = tid; e < a.K * 16; e += 256) {
        const double lp = lds_lp[e & 15];
        lds_t[e] = (lp == -INFINITY) ? 0.0 : exp(lds_t[e] - lp);   // responsibility r_k
    }
    __syncthreads();
    // z_k = y_k P_k^T: A = y (lane: candidate cl, k-step element j = 4 kk + g), B = transposed fragments
    mg_f64x4 gacc[JTM];
#pragma unroll
    for (int it = 0; it < JTM; it++) gacc[it] = {0.0, 0.0, 0.0, 0.0};
    for (int k = wave; k < a.K; k += 4) {
        double ya[KK];
#pragma unroll
        for (int kk = 0; kk < KK; kk++) ya[kk] = lds_y[(k * 16 + cl) * YS + 4 * kk + g];
        mg_f64x4 z[JTM];
#pragma unroll
        for (int it = 0; it < JTM; it++) {
            const int itc = it < a.JT ? it : a.JT - 1;
            const double *pp = PTpack + (((size_t)k * a.JT + itc) * KK) * 64 + lane;
            z[it] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KK; kk++)
                if (kk >= 4 * it) z[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[kk], pp[kk * 64], z[it], 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < JTM; it++)
#pragma unroll
            for (int r = 0; r < 4; r++) gacc[it][r] = fma(lds_t[k * 16 + g + 4 * r], z[it][r], gacc[it][r]);
    }
#pragma unroll
    for (int it = 0; it < JTM; it++)
#pragma unroll
        for (int r = 0; r < 4; r++) lds_g[(wave * 16 + g + 4 * r) * (JTM * 16) + 16 * it + cl] = gacc[it][r];
    __syncthreads();
    for (int e = tid; e < 16 * a.L; e += 256) {
        const int c = e / a.L, i = e - c * a.L;
        if (c >= ncand) continue;
        double v = 0.0;
        for (int w = 0; w < 4; w++) v += lds_g[(w * 16 + c) * (JTM * 16) + i];
        out[(b0 + c) * a.L + i] = (exp(lds_lp[c]) == 0.0) ? 1.0 : v;   // the reference: denominator == 0 -> ones
    }
}

template <int KK>
static int mg_launch_gmm_jac_mfma_kk(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, double *out) {
    mg_gmm_mfma_args a;
    a.B = B; a.ld = ld; a.K = p->K; a.L = p->Lg; a.JT = (p->Lg + 15) / 16;
    constexpr int JTM = (KK + 3) / 4;
    const int64_t grid = (B + 15) / 16;
    const size_t lds = ((size_t)p->K * 16 * (JTM * 16 + 1) + (size_t)p->K * 16 + 16 + (size_t)4 * 16 * JTM * 16) * 8;
    if (lds > 150 * 1024 || grid > 0x7fffffff) return MG_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_jac_mfma_kernel<KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_jac_mfma_kernel<KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipStream_t st = p->ctx->stream;
    if (xdt == MG_F64) hipLaunchKernelGGL((mg_gmm_jac_mfma_kernel<KK, true>), dim3((int)grid), dim3(256), lds, st, p->d_gPpack, p->d_gPTpack, p->d_gmPpad, p->d_gconst, x, out, a);
    else hipLaunchKernelGGL((mg_gmm_jac_mfma_kernel<KK, false>), dim3((int)grid), dim3(256), lds, st, p->d_gPpack, p->d_gPTpack, p->d_gmPpad, p->d_gconst, x, out, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

int mg_launch_gmm_jac(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, double *out) {
    if (p->d_gPTpack) {
        int rc = MG_ERR_UNSUPPORTED;
        switch (p->KKg) {
            case 2: rc = mg_launch_gmm_jac_mfma_kk<2>(p, x, xdt, B, ld, out); break;
            case 4: rc = mg_launch_gmm_jac_mfma_kk<4>(p, x, xdt, B, ld, out); break;
            case 6: rc = mg_launch_gmm_jac_mfma_kk<6>(p, x, xdt, B, ld, out); break;
            case 8: rc = mg_launch_gmm_jac_mfma_kk<8>(p, x, xdt, B, ld, out); break;
            case 10: rc = mg_launch_gmm_jac_mfma_kk<10>(p, x, xdt, B, ld, out); break;
            case 12: rc = mg_launch_gmm_jac_mfma_kk<12>(p, x, xdt, B, ld, out); break;
            case 14: rc = mg_launch_gmm_jac_mfma_kk<14>(p, x, xdt, B, ld, out); break;
            case 16: rc = mg_launch_gmm_jac_mfma_kk<16>(p, x, xdt, B, ld, out); break;
            default: break;
        }
        if (rc != MG_ERR_UNSUPPORTED) return rc;   // too many components for LDS: the VALU kernel below
    }
    mg_gmm_args a;
    a.P = p->d_gP; a.mP = p->d_gmP; a.cst = p->d_gconst; a.x = x; a.out = out; a.B = B; a.ld = ld; a.K = p->K; a.L = p->Lg;
    const int64_t grid = (B + MG_JAC_CANDS - 1) / MG_JAC_CANDS;
    const size_t lds = ((size_t)MG_JAC_CANDS * p->Lg * (1 + p->K) + (size_t)p->K * MG_JAC_CANDS + MG_JAC_CANDS + (size_t)p->Lg * (p->Lg + 1)) * 8;
    if (lds > 150 * 1024 || grid > 0x7fffffff || MG_JAC_CANDS * p->Lg > 256 * MG_JAC_ITEMS) {
        mg_set_error("mg_gmm_log_prob_jac: n_components %d x n_gmm %d too large for the LDS-staged kernel", p->Lg, p->K);
        return MG_ERR_UNSUPPORTED;
    }
    if (lds > 64 * 1024) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_jac_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_gmm_jac_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipStream_t st = p->ctx->stream;
    if (xdt == MG_F64) hipLaunchKernelGGL((mg_gmm_jac_kernel<true>), dim3((int)grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((mg_gmm_jac_kernel<false>), dim3((int)grid), dim3(256), lds, st, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}


// MFMA sampler (n_components <= 64): a wave owns 16 consecutive rows of ONE component (tiles never straddle
// components: the host passes tile prefix sums).  The standard normals of the tile come from the same Philox
// counters as in the VALU kernel (row, group of four) -> LDS; x = mu + z L^T is KK chained v_mfma_f64_16x16x4_f64 per
// 16-column tile with C-in = mu and the lower-triangular factor's transposed fragments (k-steps beyond the diagonal
// are skipped; zero entries inside the diagonal block add exact zeros), i.e. the VALU kernel's ascending fma chain:
// same seed, same rows, same bits.
// row and tile prefix sums as a kernel argument (n_gmm <= 16): no upload, no synchronisation per call
struct mg_cum_arg {
    int64_t v[2 * (MG_SAMPLE_ARG_K + 1)];
};

template <int KK, bool X_F64, bool CUM_ARG>
__global__ __launch_bounds__(256) void mg_gmm_sample_mfma_kernel(const double *__restrict__ cpack,     // [K][JT][KK][64]
                                                                const double *__restrict__ meanpad,   // [K][JT*16]
                                                                const int64_t *__restrict__ cum_dev,  // [K+1] rows, then [K+1] tiles
                                                                const mg_cum_arg cum_arg,
                                                                void *__restrict__ x, int32_t *__restrict__ comp,
                                                                const int64_t n_tiles, const int64_t ld, const uint64_t seed,
                                                                const int K, const int L, const int JT,
                                                                const int64_t tile0, const int64_t row_lo, const int64_t row_hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int JTM = (KK + 3) / 4;
    constexpr int ZS = 4 * KK + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;
    mg_lds_f64 *zt = (mg_lds_f64 *)smem + wave * 16 * ZS;   // [16][ZS] standard normals of this wave's tile
    const int64_t t = tile0 + (int64_t)blockIdx.x * 4 + wave;   // tiles [tile0, n_tiles): the ones that hold rows [row_lo, row_hi)
    if (t >= n_tiles) return;
    const int64_t *cum = CUM_ARG ? cum_arg.v : cum_dev;
    const int64_t *tcum = cum + K + 1;
    int c = 0;
    while (c + 1 < K && t >= tcum[c + 1]) c++;
    const int64_t row0 = cum[c] + (t - tcum[c]) * 16;
    const int nrow = (int)((cum[c + 1] - row0) < 16 ? (cum[c + 1] - row0) : 16);
    for (int e = lane; e < 16 * KK; e += 64) {
        const int r = e & 15, q = e >> 4;
        const int64_t b = row0 + r;
        double z4[4];
        mg_normal4(b, q, seed, z4);
#pragma unroll
        for (int j = 0; j < 4; j++) zt[r * ZS + 4 * q + j] = z4[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double za[KK];
#pragma unroll
    for (int kk = 0; kk < KK; kk++) za[kk] = (4 * kk + g < L) ? zt[cl * ZS + 4 * kk + g] : 0.0;
#pragma unroll
    for (int it = 0; it < JTM; it++) {
        if (it < JT) {
            const double *cp = cpack + (((size_t)c * JT + it) * KK) * 64 + lane;
            const double m = meanpad[((size_t)c * JT + it) * 16 + cl];
            mg_f64x4 acc = {m, m, m, m};
#pragma unroll
            for (int kk = 0; kk < KK; kk++)
                if (kk < 4 * (it + 1)) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(za[kk], cp[kk * 64], acc, 0, 0, 0);
            const int i = 16 * it + cl;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = g + 4 * r;
                const int64_t gr = row0 + row;
                if (row < nrow && i < L && gr >= row_lo && gr < row_hi) {
                    if (X_F64) ((double *)x)[(gr - row_lo) * ld + i] = acc[r];
                    else ((float *)x)[(gr - row_lo) * ld + i] = (float)acc[r];
                }
            }
        }
    }
    if (comp && lane < nrow && row0 + lane >= row_lo && row0 + lane < row_hi) comp[row0 + lane - row_lo] = c;
}

template <int KK>
static int mg_launch_gmm_sample_mfma_kk(mg_primitive *p, const int64_t *cum_dev, const int64_t *cum_host, int64_t n_tiles, uint64_t seed,
                                        void *x, int xdt, int64_t ld, int32_t *comp, int64_t tile0, int64_t row_lo, int64_t row_hi) {
    const int64_t grid = (n_tiles - tile0 + 3) / 4;
    if (grid <= 0) return MG_OK;
    if (grid > 0x7fffffff) return MG_ERR_UNSUPPORTED;
    const size_t lds = (size_t)4 * 16 * (4 * KK + 1) * 8;
    hipStream_t st = p->ctx->stream;
    const int JT = (p->Lg + 15) / 16;
    mg_cum_arg ca;
    memset(&ca, 0, sizeof(ca));
    if (cum_host) {
        memcpy(ca.v, cum_host, sizeof(int64_t) * 2 * (p->K + 1));
        if (xdt == MG_F64) hipLaunchKernelGGL((mg_gmm_sample_mfma_kernel<KK, true, true>), dim3((int)grid), dim3(256), lds, st, p->d_gcholpack, p->d_gmeanpad, nullptr, ca, x, comp, n_tiles, ld, seed, p->K, p->Lg, JT, tile0, row_lo, row_hi);
        else hipLaunchKernelGGL((mg_gmm_sample_mfma_kernel<KK, false, true>), dim3((int)grid), dim3(256), lds, st, p->d_gcholpack, p->d_gmeanpad, nullptr, ca, x, comp, n_tiles, ld, seed, p->K, p->Lg, JT, tile0, row_lo, row_hi);
    } else {
        if (xdt == MG_F64) hipLaunchKernelGGL((mg_gmm_sample_mfma_kernel<KK, true, false>), dim3((int)grid), dim3(256), lds, st, p->d_gcholpack, p->d_gmeanpad, cum_dev, ca, x, comp, n_tiles, ld, seed, p->K, p->Lg, JT, tile0, row_lo, row_hi);
        else hipLaunchKernelGGL((mg_gmm_sample_mfma_kernel<KK, false, false>), dim3((int)grid), dim3(256), lds, st, p->d_gcholpack, p->d_gmeanpad, cum_dev, ca, x, comp, n_tiles, ld, seed, p->K, p->Lg, JT, tile0, row_lo, row_hi);
    }
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// rows [row_lo, row_hi) of the draw of n rows; tiles [tile0, n_tiles) are the ones that hold them
int mg_launch_gmm_sample(mg_primitive *p, int64_t n, const int64_t *cum_dev, const int64_t *cum_host, int64_t n_tiles, uint64_t seed, void *x, int xdt, int64_t ld, int32_t *comp,
                         int64_t tile0, int64_t row_lo, int64_t row_hi) {
    if (p->d_gcholpack && (cum_host || cum_dev) && !p->ctx->opt[MG_OPT_FORCE_VALU_SAMPLE]) {
        switch (p->KKg) {
            case 2: return mg_launch_gmm_sample_mfma_kk<2>(p, cum_dev, cum_host, n_tiles, seed, x, xdt, ld, comp, tile0, row_lo, row_hi);
            case 4: return mg_launch_gmm_sample_mfma_kk<4>(p, cum_dev, cum_host, n_tiles, seed, x, xdt, ld, comp, tile0, row_lo, row_hi);
            case 6: return mg_launch_gmm_sample_mfma_kk<6>(p, cum_dev, cum_host, n_tiles, seed, x, xdt, ld, comp, tile0, row_lo, row_hi);
            case 8: return mg_launch_gmm_sample_mfma_kk<8>(p, cum_dev, cum_host, n_tiles, seed, x, xdt, ld, comp, tile0, row_lo, row_hi);
            case 10: return mg_launch_gmm_sample_mfma_kk<10>(p, cum_dev, cum_host, n_tiles, seed, x, xdt, ld, comp, tile0, row_lo, row_hi);
            case 12: return mg_launch_gmm_sample_mfma_kk<12>(p, cum_dev, cum_host, n_tiles, seed, x, xdt, ld, comp, tile0, row_lo, row_hi);
            case 14: return mg_launch_gmm_sample_mfma_kk<14>(p, cum_dev, cum_host, n_tiles, seed, x, xdt, ld, comp, tile0, row_lo, row_hi);
            case 16: return mg_launch_gmm_sample_mfma_kk<16>(p, cum_dev, cum_host, n_tiles, seed, x, xdt, ld, comp, tile0, row_lo, row_hi);
            default: break;
        }
    }
    return mg_launch_gmm_sample_valu(p, n, cum_dev, seed, x, xdt, ld, comp, row_lo, row_hi);
}

bool mg_gmm_sample_takes_host_prefix(const mg_primitive *p) {
    return p->d_gcholpack != nullptr && p->KKg > 0 && p->K <= MG_SAMPLE_ARG_K && !p->ctx->opt[MG_OPT_FORCE_VALU_SAMPLE];
}


// MotionPrimitive._back_transform_gamma_to_canonical_time_function (reference motion_primitive.py:289-302): one thread per
// sample walks the canonical frames, increment = exp(mean_t(i) + phi(i) . gamma) with the dot product as an ascending fma chain
__global__ __launch_bounds__(128) void mg_time_function_kernel(const double *__restrict__ tphi, const double *__restrict__ tmean,
                                                               const void *__restrict__ gamma, int gamma_f64, int64_t B, int64_t ld,
                                                               int F, int Lt, double *__restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double g[16];
    for (int l = 0; l < Lt && l < 16; l++)
        g[l] = gamma_f64 ? ((const double *)gamma)[b * ld + l] : (double)((const float *)gamma)[b * ld + l];
    double acc = 0.0;
    for (int i = 0; i < F; i++) {
        double e = tmean[i];
        for (int l = 0; l < Lt; l++) {
            const double gl = l < 16 ? g[l] : (gamma_f64 ? ((const double *)gamma)[b * ld + l] : (double)((const float *)gamma)[b * ld + l]);
            e = fma(tphi[(size_t)i * Lt + l], gl, e);
        }
        acc += exp(e);
        out[b * F + i] = acc - 1.0;
    }
}

int mg_launch_time_function(mg_primitive *p, const void *gamma, int gdt, int64_t B, int64_t ld, double *out) {
    const int grid = (int)((B + 127) / 128);
    hipLaunchKernelGGL(mg_time_function_kernel, dim3(grid), dim3(128), 0, p->ctx->stream, p->d_tphi, p->d_tmean, gamma, gdt == MG_F64 ? 1 : 0,
                       B, ld, p->F, p->Lt, out);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// Device-side pieces of the MFMA log-likelihood, shared by the stand-alone kernel (mg_gmm.hip)
// and by the fused step kernel (mg_frames_ws.hip, mg_frames_cs.hip) so that both produce the same bits.
// log p(x) = logsumexp_k [ cst_k - 0.5 |x P_k - mu_k P_k|^2 ]   (sklearn score_samples; reference
// morphablegraphs/motion_model/motion_primitive.py:126-144).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

typedef double mg_f64x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) double mg_lds_f64;   // explicit LDS pointers: ds_* ops, 32-bit addresses
typedef __attribute__((address_space(3))) float mg_lds_f32;

// A/B fragments of a 16-candidate latent tile for the 16x16x4 MFMAs: lane l supplies
// element [candidate = l & 15][k = 4*kk + (l >> 4)], 0 outside the tile / the latent dimension.
// Kept in the latents' own type (float32 latents cost half the registers) and widened at the use.
//  * every load is unconditional on a clamped in-bounds address and the zero is a select: a load under a lane
//    condition compiles to branch + load + wait, i.e. KK serialized memory round trips per fragment;
//  * KK = ceil(L/4) rounded up to even, so only the last two k-steps can reach past L: the others are
//    immediate offsets from one row pointer (one address register pair per tile).
template <bool X_F64> struct mg_gmm_xt { typedef double type; };
template <> struct mg_gmm_xt<false> { typedef float type; };

template <int KK, bool X_F64>
__device__ __forceinline__ void mg_gmm_load_x(typename mg_gmm_xt<X_F64>::type (&xf)[KK], const void *x, int64_t b0, int ncand,
                                              int64_t ld, int L, int cl, int g) {
    typedef typename mg_gmm_xt<X_F64>::type T;
    const int c = cl < ncand ? cl : ncand - 1;
    const bool rok = cl < ncand;
    const T *row = (const T *)x + (b0 + c) * ld;
    const T *rowg = row + g;
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
        if (kk < KK - 2) {
            const T v = rowg[4 * kk];
            xf[kk] = rok ? v : (T)0;
        } else {
            const int k = 4 * kk + g;
            const T v = row[k < L ? k : L - 1];
            xf[kk] = (rok && k < L) ? v : (T)0;
        }
    }
}

// One mixture component held in registers: B fragments of P_k (upper triangular: column tile jt needs only
// the k-steps < 4 (jt + 1)), C-in = -mu_k P_k, the component's constant.
template <int KK>
struct mg_gmm_frag {
    static constexpr int JTM = (KK + 3) / 4;   // column tiles of 16 for n_components <= 4 KK
    double pf[JTM][KK];
    double c0[JTM];
    double cst;
};

template <int KK>
__device__ __forceinline__ void mg_gmm_load_component(mg_gmm_frag<KK> &f,
                                                      const double *__restrict__ Ppack,  // [K][JT][KK][64], then the same in pairs per lane [K][JT][KK/2][64][2]
                                                      const double *__restrict__ mP,     // [K][JT*16]
                                                      const double *__restrict__ cst,    // [K]
                                                      int k, int JT, int lane, int cl, int K) {
    static_assert(KK % 2 == 0, "k-steps come in pairs");
    typedef double mg_f64x2 __attribute__((ext_vector_type(2)));
    constexpr int JTM = mg_gmm_frag<KK>::JTM;
    // the fragments of P from the paired copy behind the image (mg_host.hip): one 16-byte load per two k-steps, half the load instructions
    const mg_f64x2 *P2 = (const mg_f64x2 *)(Ppack + (size_t)K * JT * KK * 64);
#pragma unroll
    for (int jt = 0; jt < JTM; jt++) {
        const int jtc = jt < JT ? jt : JT - 1;
        const mg_f64x2 *pp = P2 + (((size_t)k * JT + jtc) * (KK / 2)) * 64 + lane;
#pragma unroll
        for (int q = 0; q < KK / 2; q++)
            if (2 * q < 4 * (jt + 1)) {
                const mg_f64x2 v = pp[q * 64];
                f.pf[jt][2 * q] = v[0];
                f.pf[jt][2 * q + 1] = v[1];
            }
        f.c0[jt] = -mP[((size_t)k * JT + jtc) * 16 + cl];
    }
    f.cst = cst[k];
}

// Sum over the 16 lanes of a row group, every lane ending with the total: the butterfly v += v[lane ^ 1], ^ 2, ^ 4, ^ 8 -- on the
// VALU's data-parallel-primitive path (quad permutes, then the half-row and row mirrors: after the first two steps the four
// lanes of a quad hold the same value, after the third the eight of a half row, so the mirrored lane holds what the xor
// partner holds: the same additions, the same bits as the shuffle form) instead of eight ds_bpermute per step pair.
template <int CTRL>
__device__ __forceinline__ double mg_dpp_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double mg_row16_sum(double v) {
    v += mg_dpp_f64<0xB1>(v);    // quad_perm [1, 0, 3, 2]: lane ^ 1
    v += mg_dpp_f64<0x4E>(v);    // quad_perm [2, 3, 0, 1]: lane ^ 2
    v += mg_dpp_f64<0x141>(v);   // row_half_mirror: the other quad of the half row
    v += mg_dpp_f64<0x140>(v);   // row_mirror: the other half row
    return v;
}

// The tail of a component: y = the JT accumulators (x P_k - mu_k P_k), squares summed over the column tiles in tile order,
// Mahalanobis term finished by a butterfly over the 16 lanes of a candidate row group, terms[k*16 + cand] = cst_k - 0.5 |y|^2.
// Shared by every kernel that evaluates the mixture, so that they all produce the same bits.
template <int JTM>
__device__ __forceinline__ void mg_gmm_finish_component(const mg_f64x4 (&acc)[JTM], int JT, double cst, int k, mg_lds_f64 *terms, int cl, int g) {
    double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int jt = 0; jt < JTM; jt++)
        if (jt < JT) {
#pragma unroll
            for (int r = 0; r < 4; r++) part[r] = fma(acc[jt][r], acc[jt][r], part[r]);
        }
    // C/D layout: col = lane & 15, row (candidate) = (lane >> 4) + 4*reg: reduce over the 16 columns
#pragma unroll
    for (int r = 0; r < 4; r++) part[r] = mg_row16_sum(part[r]);
    if (cl == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) terms[k * 16 + g + 4 * r] = cst - 0.5 * part[r];
    }
}

// The component applied to a 16-candidate latent tile by one wave:
// terms[k*16 + cand] = cst_k - 0.5 |x P_k - mu_k P_k|^2, the column tiles' accumulator chains interleaved.
// WIDEN_AT_USE: the latent is widened again at every call (an opaque copy first), so that a caller that applies several components to
// several tiles keeps its tiles in float32 registers instead of the compiler's hoisted float64 copies (the same conversions, the same bits).
template <int KK, typename T, bool WIDEN_AT_USE = false>
__device__ __forceinline__ void mg_gmm_apply_component(const mg_gmm_frag<KK> &f, int k, int JT, const T (&xf)[KK],
                                                       mg_lds_f64 *terms, int cl, int g) {
    constexpr int JTM = mg_gmm_frag<KK>::JTM;
    mg_f64x4 acc[JTM];
#pragma unroll
    for (int jt = 0; jt < JTM; jt++) acc[jt] = {f.c0[jt], f.c0[jt], f.c0[jt], f.c0[jt]};
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
        T xv = xf[kk];
        if constexpr (WIDEN_AT_USE) asm volatile("" : "+v"(xv));
        const double xd = (double)xv;
#pragma unroll
        for (int jt = 0; jt < JTM; jt++)
            if (kk < 4 * (jt + 1)) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, f.pf[jt][kk], acc[jt], 0, 0, 0);
    }
    mg_gmm_finish_component<JTM>(acc, JT, f.cst, k, terms, cl, g);
}

// The same with the latent tile's float32 A fragments in LDS ([KK][64], lane-major: what mg_gmm_load_x returns, stored as it is) and the
// components' C-in rows and constants in LDS as well (mpl = mP as it lies in memory, [K][JT*16]; cstl [K]): between two components a wave
// holds nothing but the fragments of P.  Same conversions, same MFMA order: the same bits.
// ... from the paired copy behind Ppack ([K][JT][KK/2][64][2], mg_host.hip): two blocks per 16-byte load
template <int KK>
__device__ __forceinline__ void mg_gmm_load_pf2(mg_gmm_frag<KK> &f, const double *__restrict__ Ppack, int K, int k, int JT, int lane) {
    static_assert(KK % 2 == 0, "k-steps come in pairs");
    typedef double mg_f64x2 __attribute__((ext_vector_type(2)));
    constexpr int JTM = mg_gmm_frag<KK>::JTM;
    const mg_f64x2 *P2 = (const mg_f64x2 *)(Ppack + (size_t)K * JT * KK * 64);
#pragma unroll
    for (int jt = 0; jt < JTM; jt++) {
        const int jtc = jt < JT ? jt : JT - 1;
        const mg_f64x2 *pp = P2 + (((size_t)k * JT + jtc) * (KK / 2)) * 64 + lane;
#pragma unroll
        for (int q = 0; q < KK / 2; q++)
            if (2 * q < 4 * (jt + 1)) {
                const mg_f64x2 v = pp[q * 64];
                f.pf[jt][2 * q] = v[0];
                f.pf[jt][2 * q + 1] = v[1];
            }
    }
}
template <int KK>
__device__ __forceinline__ void mg_gmm_load_pf(mg_gmm_frag<KK> &f, const double *__restrict__ Ppack, int k, int JT, int lane) {
    constexpr int JTM = mg_gmm_frag<KK>::JTM;
#pragma unroll
    for (int jt = 0; jt < JTM; jt++) {
        const int jtc = jt < JT ? jt : JT - 1;
        const double *pp = Ppack + (((size_t)k * JT + jtc) * KK) * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < KK; kk++)
            if (kk < 4 * (jt + 1)) f.pf[jt][kk] = pp[kk * 64];
    }
}
template <int KK>
__device__ __forceinline__ void mg_gmm_apply_component_ldsx(const mg_gmm_frag<KK> &f, int k, int JT, const mg_lds_f32 *x, int lane,
                                                            const mg_lds_f64 *mpl, const mg_lds_f64 *cstl, mg_lds_f64 *terms, int cl, int g) {
    constexpr int JTM = mg_gmm_frag<KK>::JTM;
    mg_f64x4 acc[JTM];
    float xv[KK];
#pragma unroll
    for (int kk = 0; kk < KK; kk++) xv[kk] = x[kk * 64 + lane];
#pragma unroll
    for (int jt = 0; jt < JTM; jt++) {
        const double c0 = -mpl[(k * JT + (jt < JT ? jt : JT - 1)) * 16 + cl];
        acc[jt] = {c0, c0, c0, c0};
    }
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
        const double xd = (double)xv[kk];
#pragma unroll
        for (int jt = 0; jt < JTM; jt++)
            if (kk < 4 * (jt + 1)) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, f.pf[jt][kk], acc[jt], 0, 0, 0);
    }
    mg_gmm_finish_component<JTM>(acc, JT, cstl[k], k, terms, cl, g);
}

// exp(term - max over the components of the same candidate) of entry e = k*16 + cand
__device__ __forceinline__ double mg_gmm_exp_entry(const mg_lds_f64 *terms, int K, int e) {
    const int c = e & 15;
    double vmax = -INFINITY;
    for (int k = 0; k < K; k++) vmax = fmax(vmax, terms[k * 16 + c]);
    return (vmax == -INFINITY) ? 0.0 : exp(terms[e] - vmax);
}

// log-sum-exp of candidate c: the exponentials summed in component order
__device__ __forceinline__ double mg_gmm_logsumexp(const mg_lds_f64 *terms, const mg_lds_f64 *exps, int K, int c) {
    double vmax = -INFINITY;
    for (int k = 0; k < K; k++) vmax = fmax(vmax, terms[k * 16 + c]);
    if (vmax == -INFINITY) return -INFINITY;
    double acc = 0.0;
    for (int k = 0; k < K; k++) acc += exps[k * 16 + c];
    return log(acc) + vmax;
}

// Philox4x32-10 (the device sampler's counter-based generator): counter (c0..c3), key (k0, k1) -> four 32-bit draws
__device__ __forceinline__ void mg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1, uint32_t *out) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}


// Four standard normals for (row b, group of four q) of a sampler call: one Philox block, two Box-Muller pairs (float32
// transcendental units since round 4, see below; returned as float64).  u in (0, 1] for the logarithm; the angle in revolutions.  Every
// device sampler -- lane per row, MFMA tiles, the one-launch planner step -- draws through this function: same seed, same
// rows, same bits.
#ifndef MG_NORMAL_F32
#define MG_NORMAL_F32 1
#endif
__device__ __forceinline__ void mg_normal4(int64_t b, int q, uint64_t seed, double (&z)[4]) {
    uint32_t rr[4];
    mg_philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), (uint32_t)q, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rr);
    const double inv = 1.0 / 4294967296.0;
    const double u0 = ((double)rr[0] + 1.0) * inv, u1 = (double)rr[1] * inv;
    const double u2 = ((double)rr[2] + 1.0) * inv, u3 = (double)rr[3] * inv;
#ifdef MG_BM_SKIP
    z[0] = u0; z[1] = u1; z[2] = u2; z[3] = u3;
#elif MG_NORMAL_F32
    // Box-Muller on the float32 transcendental units (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32, whose argument is in revolutions:
    // no 2 pi).  The uniforms carry 32 bits, so the normals never had more than that to say; in float64 the two logarithms, two
    // roots and two sincospi per call were ~340 vector instructions and a fifth of the planner step's kernel.  Deterministic on
    // gfx950; the contract of the device sampler is distributional (include/mg_hip.h), the tests hold moments and identities.
    const float m0 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf((float)u0));   // -2 ln u = -2 ln 2 log2 u
    const float m1 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf((float)u2));
    const float a0 = (float)u1, a1 = (float)u3;
    z[0] = (double)(m0 * __builtin_amdgcn_cosf(a0)); z[1] = (double)(m0 * __builtin_amdgcn_sinf(a0));
    z[2] = (double)(m1 * __builtin_amdgcn_cosf(a1)); z[3] = (double)(m1 * __builtin_amdgcn_sinf(a1));
#else
    const double m0 = sqrt(-2.0 * log(u0)), m1 = sqrt(-2.0 * log(u2));
    double s0, c0, s1, c1;
    sincospi(2.0 * u1, &s0, &c0);
    sincospi(2.0 * u3, &s1, &c1);
    z[0] = m0 * c0; z[1] = m0 * s0; z[2] = m1 * c1; z[3] = m1 * s1;
#endif
}

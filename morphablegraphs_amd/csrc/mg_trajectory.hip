// Trajectory constraints on the device (gfx950): TrajectoryConstraint.get_residual_vector / evaluate_motion_spline for
// the ROOT joint (reference morphablegraphs/constraints/spatial_constraints/trajectory_constraint.py:79-121):
// per frame of the motion the distance from the root position to the closest point of a Catmull-Rom spline
// (splines/catmull_rom_spline.py:118-168) whose parameter is at or after the previous frame's; the constraint's error is
// the average over the frames.  The reference finds that point with scipy's L-BFGS-B started at the lower bound
// (splines/parameterized_spline.py:303-322).  Round 5: PINNED -- the reference's own function produced
// tests/golden/trajectory_closest_point.npz (with a size-1-array shim for NumPy >= 1.24, oracle/gen_golden.py), and the search here
// IS that algorithm restated for one bounded variable (mg_traj_device.h: mg_traj_closest_lbfgsb; why nothing less reproduces the
// reference is told there).  The deterministic walk of rounds 2-4 -- on the grid u_k = k / granularity forward from the bound while
// the distance falls, parabola, Newton steps: the local minimum of the first basin at or after the bound -- stays as
// MG_OPT_TRAJECTORY_SEARCH = 1: equal to the reference where the distance has one basin ahead of the bound (smooth path following),
// up to 0.9 of the parameter range away from it where it has several.
//
// One thread per candidate (the search is a chain over the frames): the candidate's root coefficient rows
// (n_basis x 3, float64 fma chains over the latents) are staged in LDS, every frame is four taps of them.
// 172 + 8 bytes per candidate: arithmetic / latency bound, a scoring kernel beside mg_score_constraints.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "mg_internal.h"
#include "mg_traj_device.h"

struct mg_traj_args {
    const double *poly, *E, *mean;
    const void *lat;
    const int32_t *i0;
    const double *w;
    double *out, *res;
    int64_t B, ld;
    int32_t T, L, NB, n_seg, G, lat_f64, accumulate, align_mode;   // align_mode 0: none, 1: previous frame (root node), 2: start pose
    double min_u, weight;
    double al[7];   // heading (x, z) or (cos, sin); landing (x, z); ref_dir (3) or height
    const double *points;   // NULL, or (B, T, 3): the positions to follow the trajectory with, given instead of derived from the
                            // candidates' root rows (any joint's track from mg_joint_positions; already aligned by the caller)
    double *res_u;          // NULL, or (B, T): the parameter the search settles on in every frame (mg_trajectory_closest_points)
    int32_t *res_n;         // NULL, or (B, T): the (f, g) evaluations every frame's search took (the reference's search only)
    double *paths;          // the reference's search on candidates' root paths: (B, T, 3) scratch the kernel writes the paths to first (a lane
                            // then searches ITS frames at its own pace, mg_traj_chain)
    int32_t search;         // 0: the reference's search (L-BFGS-B restated, mg_traj_closest_lbfgsb), 1: the monotone walk (MG_OPT_TRAJECTORY_SEARCH)
};

#define MG_TRAJ_BLOCK 64

// POLY_LDS: the target spline's segment polynomials (12 n_seg + 3 doubles) are copied behind the rows and searched from there -- the walk
// is a chain of dependent evaluations (about 17 per frame), each one a look-up of its segment: an LDS read instead of a trip to L2
template <bool POLY_LDS>
__global__ __launch_bounds__(MG_TRAJ_BLOCK) void mg_trajectory_kernel(mg_traj_args a) {
    extern __shared__ double lds[];                 // [L + rows][64]: the latents, then the root coefficient rows; [12 n_seg + 3] the polynomials
    const int tid = threadIdx.x;
    const int64_t b = (int64_t)blockIdx.x * MG_TRAJ_BLOCK + tid;
    const bool valid = b < a.B;
    const int64_t bb = valid ? b : a.B - 1;
    const int rows = a.NB * 3 + 4;
    double *ls = lds, *lc = lds + (size_t)a.L * MG_TRAJ_BLOCK;
    const double *poly = a.poly;
    if constexpr (POLY_LDS) {
        double *lp = lds + (a.points ? 0 : (size_t)(a.L + rows) * MG_TRAJ_BLOCK);
        for (int e = tid; e < a.n_seg * 12 + 3; e += MG_TRAJ_BLOCK) lp[e] = a.poly[e];
        poly = lp;
        __syncthreads();
    }
    if (!a.points) {
        for (int k = 0; k < a.L; k++)
            ls[k * MG_TRAJ_BLOCK + tid] = a.lat_f64 ? ((const double *)a.lat)[bb * a.ld + k] : (double)((const float *)a.lat)[bb * a.ld + k];
        for (int r = 0; r < rows; r++) {
            double acc = a.mean[r];
            const double *e = a.E + (size_t)r * a.L;
            for (int k = 0; k < a.L; k++) acc = fma(e[k], ls[k * MG_TRAJ_BLOCK + tid], acc);
            lc[r * MG_TRAJ_BLOCK + tid] = acc;
        }
    }
    // the candidate's aligning transform (mg_score.hip's closed form): rotation about y and an xz translation
    double ac = 1.0, as = 0.0, tx = 0.0, ty = 0.0, tz = 0.0;
    if (a.align_mode != 0) {
        if (a.align_mode == 2) {
            ac = a.al[0]; as = a.al[1]; ty = a.al[4];
        } else {
            double qw = lc[(a.NB * 3 + 0) * MG_TRAJ_BLOCK + tid], qx = lc[(a.NB * 3 + 1) * MG_TRAJ_BLOCK + tid];
            double qy = lc[(a.NB * 3 + 2) * MG_TRAJ_BLOCK + tid], qz = lc[(a.NB * 3 + 3) * MG_TRAJ_BLOCK + tid];
            const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
            qw *= inv; qx *= inv; qy *= inv; qz *= inv;
            const double rx = a.al[4], ry = a.al[5], rz = a.al[6];
            const double cx = qy * rz - qz * ry, cy = qz * rx - qx * rz, cz = qx * ry - qy * rx;
            const double dx = qy * cz - qz * cy, dz = qx * cy - qy * cx;
            double bx = rx + 2.0 * (qw * cx + dx), bz = rz + 2.0 * (qw * cz + dz);
            const double bn = 1.0 / sqrt(bx * bx + bz * bz);
            bx *= bn; bz *= bn;
            ac = a.al[0] * bx + a.al[1] * bz;
            as = a.al[0] * bz - a.al[1] * bx;
        }
        const double p0x = lc[0 * MG_TRAJ_BLOCK + tid], p0z = lc[2 * MG_TRAJ_BLOCK + tid];
        tx = a.al[2] - (ac * p0x + as * p0z);
        tz = a.al[3] - (ac * p0z - as * p0x);
    }
    const int G = a.G;
    const double invG = 1.0 / (double)G;
    double min_u = a.min_u, sum = 0.0;
    auto position = [&](int f, double *q) {        // the candidate's root position in frame f, aligned
        const int i0 = a.i0[f];
        const double *w = a.w + 4 * (size_t)f;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double v = w[0] * lc[((i0 + 0) * 3 + d) * MG_TRAJ_BLOCK + tid];
            v = fma(w[1], lc[((i0 + 1) * 3 + d) * MG_TRAJ_BLOCK + tid], v);
            v = fma(w[2], lc[((i0 + 2) * 3 + d) * MG_TRAJ_BLOCK + tid], v);
            v = fma(w[3], lc[((i0 + 3) * 3 + d) * MG_TRAJ_BLOCK + tid], v);
            q[d] = v;
        }
        if (a.align_mode != 0) {
            const double x = q[0], z = q[2];
            q[0] = ac * x + as * z + tx;
            q[2] = ac * z - as * x + tz;
            q[1] += ty;
        }
    };
    if (a.search == 0) {
        // the reference's search: every lane walks ITS frames at its own pace (mg_traj_chain); the positions come from the caller's
        // points or, for the root path, are evaluated per lane from the rows in LDS
        const double *pts = a.points ? a.points + (size_t)bb * a.T * 3 : nullptr;
        mg_traj_chain(poly, a.n_seg, valid ? a.T : 0, a.min_u,
                      [&](int f, double *q) { if (pts) { q[0] = pts[3 * f]; q[1] = pts[3 * f + 1]; q[2] = pts[3 * f + 2]; } else position(f, q); },
                      [&](int f, double dist, double u, int trips) {
                          sum += dist;
                          if (a.res) a.res[b * a.T + f] = a.weight * dist;
                          if (a.res_u) a.res_u[b * a.T + f] = u;
                          if (a.res_n) a.res_n[b * a.T + f] = trips;
                      });
    } else {
        // (given positions: the next frame's point is requested a frame ahead -- its trip to memory would otherwise head every frame's
        // chain of dependent evaluations)
        double qn[3] = {0.0, 0.0, 0.0};
        if (a.points && a.T > 0) { const double *pp = a.points + (size_t)bb * a.T * 3; qn[0] = pp[0]; qn[1] = pp[1]; qn[2] = pp[2]; }
        for (int f = 0; f < a.T; f++) {
            double q[3];
            if (a.points) {
                q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2];
                if (f + 1 < a.T) { const double *pp = a.points + ((size_t)bb * a.T + f + 1) * 3; qn[0] = pp[0]; qn[1] = pp[1]; qn[2] = pp[2]; }
            } else position(f, q);
            // (mg_traj_device.h; min_u moves to the point's parameter; every lane of the wave is here: the walk's long searches get its help)
            const double dist = mg_traj_closest_dist<true>(poly, a.n_seg, G, invG, &min_u, q);
            sum += dist;
            if (a.res && valid) a.res[b * a.T + f] = a.weight * dist;
            if (a.res_u && valid) a.res_u[b * a.T + f] = min_u;
        }
    }
    if (valid && a.out) {
        const double e = a.weight * (a.T > 0 ? sum / (double)a.T : 0.0);
        a.out[b] = a.accumulate ? a.out[b] + e : e;
    }
}

// The same scorer with MG_TRAJ_W lanes per candidate, for batches that leave most of the chip idle at one lane per candidate (4096
// candidates are 64 waves on 1024 SIMDs, each a chain of ~18 dependent float64 evaluations per frame): the lanes of a candidate split
// its coefficient rows (each row the same fma chain as above), then walk the frames together -- every lane the same q, the grid values
// of the closest-point search side by side (mg_traj_closest_dist_coop).  Same operations on the same values: the same bits.
// (W = 8 or 4: with four lanes a window holds three grid values and a walk of six steps takes three more rounds, but the Newton steps
// are repeated four times, not eight: measured, four lanes win only between ~28 000 and ~40 000 candidates in flight -- 762 us at
// 32 768 against 899 with eight and 888 with one)
#define MG_TRAJ_COOP_BLOCK 64
template <int MG_TRAJ_W>
__device__ __forceinline__ void mg_trajectory_coop_body(const mg_traj_args &a, const int64_t block, double *lds) {
    constexpr int MG_TRAJ_COOP_CANDS = MG_TRAJ_COOP_BLOCK / MG_TRAJ_W;
    // lds: [CANDS][L] latents, [CANDS][rows] root coefficient rows, [12 n_seg + 3] the polynomials
    const int tid = threadIdx.x, grp = tid / MG_TRAJ_W, sub = tid % MG_TRAJ_W;
    const int64_t b = block * MG_TRAJ_COOP_CANDS + grp;
    const bool valid = b < a.B;
    const int64_t bb = valid ? b : a.B - 1;
    const int rows = a.NB * 3 + 4;
    double *ls = lds + (size_t)grp * a.L, *lc = lds + (size_t)MG_TRAJ_COOP_CANDS * a.L + (size_t)grp * rows;
    double *lp = lds + (a.points ? 0 : (size_t)MG_TRAJ_COOP_CANDS * (a.L + rows));
    for (int e = tid; e < a.n_seg * 12 + 3; e += MG_TRAJ_COOP_BLOCK) lp[e] = a.poly[e];
    if (!a.points) {
        for (int k = sub; k < a.L; k += MG_TRAJ_W) ls[k] = a.lat_f64 ? ((const double *)a.lat)[bb * a.ld + k] : (double)((const float *)a.lat)[bb * a.ld + k];
        __syncthreads();
        for (int r = sub; r < rows; r += MG_TRAJ_W) {
            double acc = a.mean[r];
            const double *e = a.E + (size_t)r * a.L;
            for (int k = 0; k < a.L; k++) acc = fma(e[k], ls[k], acc);
            lc[r] = acc;
        }
    }
    __syncthreads();
    double ac = 1.0, as = 0.0, tx = 0.0, ty = 0.0, tz = 0.0;
    if (a.align_mode != 0) {
        if (a.align_mode == 2) {
            ac = a.al[0]; as = a.al[1]; ty = a.al[4];
        } else {
            double qw = lc[a.NB * 3 + 0], qx = lc[a.NB * 3 + 1], qy = lc[a.NB * 3 + 2], qz = lc[a.NB * 3 + 3];
            const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
            qw *= inv; qx *= inv; qy *= inv; qz *= inv;
            const double rx = a.al[4], ry = a.al[5], rz = a.al[6];
            const double cx = qy * rz - qz * ry, cy = qz * rx - qx * rz, cz = qx * ry - qy * rx;
            const double dx = qy * cz - qz * cy, dz = qx * cy - qy * cx;
            double bx = rx + 2.0 * (qw * cx + dx), bz = rz + 2.0 * (qw * cz + dz);
            const double bn = 1.0 / sqrt(bx * bx + bz * bz);
            bx *= bn; bz *= bn;
            ac = a.al[0] * bx + a.al[1] * bz;
            as = a.al[0] * bz - a.al[1] * bx;
        }
        const double p0x = lc[0], p0z = lc[2];
        tx = a.al[2] - (ac * p0x + as * p0z);
        tz = a.al[3] - (ac * p0z - as * p0x);
    }
    const int G = a.G;
    const double invG = 1.0 / (double)G;
    double min_u = a.min_u, sum = 0.0;
    // (given positions: the next frame's point is requested a frame ahead -- its trip to memory would otherwise head every frame's
    // chain of dependent evaluations)
    double qn[3] = {0.0, 0.0, 0.0};
    if (a.points && a.T > 0) { const double *pp = a.points + (size_t)bb * a.T * 3; qn[0] = pp[0]; qn[1] = pp[1]; qn[2] = pp[2]; }
    for (int f = 0; f < a.T; f++) {
        double q[3];
        if (a.points) {
            q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2];
            if (f + 1 < a.T) { const double *pp = a.points + ((size_t)bb * a.T + f + 1) * 3; qn[0] = pp[0]; qn[1] = pp[1]; qn[2] = pp[2]; }
        } else {
            const int i0 = a.i0[f];
            const double *w = a.w + 4 * (size_t)f;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                double v = w[0] * lc[(i0 + 0) * 3 + d];
                v = fma(w[1], lc[(i0 + 1) * 3 + d], v);
                v = fma(w[2], lc[(i0 + 2) * 3 + d], v);
                v = fma(w[3], lc[(i0 + 3) * 3 + d], v);
                q[d] = v;
            }
        }
        if (a.align_mode != 0) {
            const double x = q[0], z = q[2];
            q[0] = ac * x + as * z + tx;
            q[2] = ac * z - as * x + tz;
            q[1] += ty;
        }
        const double dist = mg_traj_closest_dist_coop<MG_TRAJ_W>(lp, a.n_seg, G, invG, &min_u, q);
        sum += dist;
        if (a.res && valid && sub == 0) a.res[b * a.T + f] = a.weight * dist;
        if (a.res_u && valid && sub == 0) a.res_u[b * a.T + f] = min_u;
    }
    if (valid && sub == 0 && a.out) {
        const double e = a.weight * (a.T > 0 ? sum / (double)a.T : 0.0);
        a.out[b] = a.accumulate ? a.out[b] + e : e;
    }
}
template <int W>
__global__ __launch_bounds__(MG_TRAJ_COOP_BLOCK) void mg_trajectory_coop_kernel(mg_traj_args a) {
    extern __shared__ double lds[];
    mg_trajectory_coop_body<W>(a, blockIdx.x, lds);
}

// One lane per candidate for batches that fill the chip that way (from ~32 768 candidates in flight: eight lanes per candidate repeat
// the Newton steps eight times, which is what counts once every SIMD has waves to choose from) -- WITHOUT the candidate's 97
// coefficient rows in LDS: mg_trajectory_kernel keeps them there, 50 KB per wave, three waves per CU.  A frame's position needs the four
// control points of its knot span only, and the span moves one control point at a time: a window of 4 x 3 control points lives in
// registers, the next control point (three fma chains over the latents, the statements of the staged form) is made when the span
// moves (every 5.6 frames for 'walk').  LDS: the latents (20 KB per wave) and the polynomials.  Same operations, same bits.
__device__ __forceinline__ void mg_trajectory_stream_body(const mg_traj_args &a, const int64_t block, double *lds) {
    const int tid = threadIdx.x;
    const int64_t b = block * MG_TRAJ_BLOCK + tid;
    const bool valid = b < a.B;
    const int64_t bb = valid ? b : a.B - 1;
    double *ls = lds, *lp = lds + (size_t)a.L * MG_TRAJ_BLOCK;
    for (int e = tid; e < a.n_seg * 12 + 3; e += MG_TRAJ_BLOCK) lp[e] = a.poly[e];
    for (int k = 0; k < a.L; k++)
        ls[k * MG_TRAJ_BLOCK + tid] = a.lat_f64 ? ((const double *)a.lat)[bb * a.ld + k] : (double)((const float *)a.lat)[bb * a.ld + k];
    __syncthreads();
    auto row = [&](int r) {
        double acc = a.mean[r];
        const double *e = a.E + (size_t)r * a.L;
        for (int k = 0; k < a.L; k++) acc = fma(e[k], ls[k * MG_TRAJ_BLOCK + tid], acc);
        return acc;
    };
    double ac = 1.0, as = 0.0, tx = 0.0, ty = 0.0, tz = 0.0;
    if (a.align_mode != 0) {
        if (a.align_mode == 2) {
            ac = a.al[0]; as = a.al[1]; ty = a.al[4];
        } else {
            double qw = row(a.NB * 3 + 0), qx = row(a.NB * 3 + 1), qy = row(a.NB * 3 + 2), qz = row(a.NB * 3 + 3);
            const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
            qw *= inv; qx *= inv; qy *= inv; qz *= inv;
            const double rx = a.al[4], ry = a.al[5], rz = a.al[6];
            const double cx = qy * rz - qz * ry, cy = qz * rx - qx * rz, cz = qx * ry - qy * rx;
            const double dx = qy * cz - qz * cy, dz = qx * cy - qy * cx;
            double bx = rx + 2.0 * (qw * cx + dx), bz = rz + 2.0 * (qw * cz + dz);
            const double bn = 1.0 / sqrt(bx * bx + bz * bz);
            bx *= bn; bz *= bn;
            ac = a.al[0] * bx + a.al[1] * bz;
            as = a.al[0] * bz - a.al[1] * bx;
        }
        const double p0x = row(0), p0z = row(2);
        tx = a.al[2] - (ac * p0x + as * p0z);
        tz = a.al[3] - (ac * p0z - as * p0x);
    }
    const int G = a.G;
    const double invG = 1.0 / (double)G;
    double min_u = a.min_u, sum = 0.0;
    double cw[4][3];                       // control points cur .. cur + 3 of the three root channels
    int cur = INT_MIN;
    auto position = [&](int f, double *q) {  // frame f's aligned root position (called with f ascending, the same f in every lane)
        const int i0 = a.i0[f];             // (the same for every lane: the branches below are uniform)
        if (i0 != cur) {
            if (cur != INT_MIN && i0 == cur + 1) {
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int d = 0; d < 3; d++) cw[j][d] = cw[j + 1][d];
#pragma unroll
                for (int d = 0; d < 3; d++) cw[3][d] = row((i0 + 3) * 3 + d);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int d = 0; d < 3; d++) cw[j][d] = row((i0 + j) * 3 + d);
            }
            cur = i0;
        }
        const double *w = a.w + 4 * (size_t)f;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double v = w[0] * cw[0][d];
            v = fma(w[1], cw[1][d], v);
            v = fma(w[2], cw[2][d], v);
            v = fma(w[3], cw[3][d], v);
            q[d] = v;
        }
        if (a.align_mode != 0) {
            const double x = q[0], z = q[2];
            q[0] = ac * x + as * z + tx;
            q[2] = ac * z - as * x + tz;
            q[1] += ty;
        }
    };
    if (a.search == 0) {
        // the reference's search: first the whole path, frame by frame with the window in registers (the lanes in step), into the
        // caller's scratch; then every lane searches ITS frames at its own pace (a lane reads back what it wrote itself)
        double *mine = a.paths + (size_t)bb * a.T * 3;
        if (valid)
            for (int f = 0; f < a.T; f++) {
                double q[3];
                position(f, q);
                mine[3 * f] = q[0]; mine[3 * f + 1] = q[1]; mine[3 * f + 2] = q[2];
            }
        mg_traj_chain(lp, a.n_seg, valid ? a.T : 0, a.min_u,
                      [&](int f, double *q) { q[0] = mine[3 * f]; q[1] = mine[3 * f + 1]; q[2] = mine[3 * f + 2]; },
                      [&](int f, double dist, double u, int trips) {
                          sum += dist;
                          if (a.res) a.res[b * a.T + f] = a.weight * dist;
                      });
    } else {
        for (int f = 0; f < a.T; f++) {
            double q[3];
            position(f, q);
            const double dist = mg_traj_closest_dist<true>(lp, a.n_seg, G, invG, &min_u, q);   // (every lane of the wave is here: the help is legal)
            sum += dist;
            if (a.res && valid) a.res[b * a.T + f] = a.weight * dist;
        }
    }
    if (valid) {
        const double e = a.weight * (a.T > 0 ? sum / (double)a.T : 0.0);
        a.out[b] = a.accumulate ? a.out[b] + e : e;
    }
}
__global__ __launch_bounds__(MG_TRAJ_BLOCK) void mg_trajectory_stream_kernel(mg_traj_args a) {
    extern __shared__ double lds[];
    mg_trajectory_stream_body(a, blockIdx.x, lds);
}
// Several scorers in ONE launch (mg_score_trajectories): a planner step scores every option's candidates against the option's own
// trajectory -- 4096 candidates at eight lanes each fill a quarter of the chip and take 0.4 ms, sixteen such launches one after the
// other 7 ms; side by side they take what the chip's four quarters take.
#define MG_TRAJ_MULTI_MAX 16
struct mg_traj_multi { int32_t n; int32_t wg0[MG_TRAJ_MULTI_MAX + 1]; mg_traj_args a[MG_TRAJ_MULTI_MAX]; };
template <int W>
__global__ __launch_bounds__(MG_TRAJ_COOP_BLOCK) void mg_trajectory_coop_multi_kernel(const mg_traj_multi m) {
    extern __shared__ double lds[];
    int k = 0;
    while (k + 1 < m.n && (int)blockIdx.x >= m.wg0[k + 1]) k++;
    mg_trajectory_coop_body<W>(m.a[k], (int64_t)blockIdx.x - m.wg0[k], lds);
}
__global__ __launch_bounds__(MG_TRAJ_BLOCK) void mg_trajectory_stream_multi_kernel(const mg_traj_multi m) {
    extern __shared__ double lds[];
    int k = 0;
    while (k + 1 < m.n && (int)blockIdx.x >= m.wg0[k + 1]) k++;
    mg_trajectory_stream_body(m.a[k], (int64_t)blockIdx.x - m.wg0[k], lds);
}
// Lanes per candidate for `total` candidates in flight (measured: tools/probes/trajectory_lanes.py): eight while most SIMDs would idle
// otherwise, four in between, one (the streaming kernel) once one lane per candidate fills the chip.  MG_OPT_TRAJECTORY_LANES forces.
#define MG_TRAJ_COOP_MAX_B 65536       // (no batch above this takes several lanes per candidate, whatever is forced)
#define MG_TRAJ_W8_MAX_TOTAL 28672
#define MG_TRAJ_W4_MAX_TOTAL 40960
static int mg_traj_lanes(const mg_context *ctx, int64_t B, int64_t total) {
    const int opt = ctx->opt[MG_OPT_TRAJECTORY_LANES];
    if (ctx->opt[MG_OPT_TRAJECTORY_SEARCH] != 1) return 1;    // the reference's search is one chain of evaluations: nothing for more lanes to do
    if (opt == 1 || B > MG_TRAJ_COOP_MAX_B) return 1;
    if (opt == 8 || opt == 4) return opt;
    return total <= MG_TRAJ_W8_MAX_TOTAL ? 8 : (total <= MG_TRAJ_W4_MAX_TOTAL ? 4 : 1);
}

// the context's buffer for the candidates' root paths (the reference's search in the streaming kernels): grown on demand, stream ordered
static int mg_traj_paths(mg_context *ctx, size_t bytes, double **out) {
    if (ctx->traj_paths_bytes < bytes) {
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));     // a launch may still be reading the old one
        if (ctx->traj_paths) (void)hipFree(ctx->traj_paths);
        ctx->traj_paths = nullptr; ctx->traj_paths_bytes = 0;
        const size_t want = std::max<size_t>(bytes, (size_t)16 << 20);
        MG_HIP_CHECK(hipMalloc(&ctx->traj_paths, want));
        ctx->traj_paths_bytes = want;
    }
    *out = (double *)ctx->traj_paths;
    return MG_OK;
}

template <typename T>
static int mg_traj_upload(const std::vector<T> &h, T **d) {
    *d = nullptr;
    MG_HIP_CHECK(hipMalloc((void **)d, std::max<size_t>(h.size() * sizeof(T), 16)));
    if (!h.empty()) MG_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return MG_OK;
}

extern "C" void mg_trajectory_destroy(mg_trajectory *t) {
    if (!t) return;
    if (t->prim) { (void)hipSetDevice(t->prim->ctx->device); (void)hipStreamSynchronize(t->prim->ctx->stream); }
    (void)hipFree(t->d_poly);
    (void)hipFree(t->d_E);
    (void)hipFree(t->d_mean);
    (void)hipFree(t->d_arc);
    delete t;
}

extern "C" int mg_trajectory_create(mg_primitive *p, const double *cp, int32_t n_points, int32_t granularity, mg_trajectory **out) {
    if (!p || !cp || !out || n_points < 2 || granularity < 2 || p->D < 3) {
        mg_set_error("mg_trajectory_create: needs a primitive with root channels, >= 2 control points (x, y, z) and a granularity >= 2");
        return MG_ERR_INVALID_ARGUMENT;
    }
    *out = nullptr;
    for (int i = 0; i < 3 * n_points; i++)
        if (!std::isfinite(cp[i])) { mg_set_error("mg_trajectory_create: control point %d is not finite", i / 3); return MG_ERR_INVALID_ARGUMENT; }
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    mg_trajectory *t = new (std::nothrow) mg_trajectory();
    if (!t) return MG_ERR_OUT_OF_MEMORY;
    t->prim = p;
    t->n_seg = n_points - 1;
    t->granularity = granularity;
    // catmull_rom_spline.py:66-71: control points padded to [P0] + P + [Pn, Pn]; segment s (1-based) uses padded[s-1 .. s+2]
    std::vector<double> pad((size_t)(n_points + 3) * 3);
    for (int d = 0; d < 3; d++) {
        pad[d] = cp[d];
        for (int i = 0; i < n_points; i++) pad[(size_t)(i + 1) * 3 + d] = cp[(size_t)i * 3 + d];
        pad[(size_t)(n_points + 1) * 3 + d] = pad[(size_t)(n_points + 2) * 3 + d] = cp[(size_t)(n_points - 1) * 3 + d];
    }
    static const double M[4][4] = {{-1.0, 3.0, -3.0, 1.0}, {2.0, -5.0, 4.0, -1.0}, {-1.0, 0.0, 1.0, 0.0}, {0.0, 2.0, 0.0, 0.0}};
    std::vector<double> poly((size_t)t->n_seg * 12 + 3);
    for (int s = 0; s < t->n_seg; s++)           // 0-based segment s = the reference's segment index s + 1
        for (int r = 0; r < 4; r++)
            for (int d = 0; d < 3; d++) {
                double v = 0.0;
                for (int j = 0; j < 4; j++) v += M[r][j] * pad[(size_t)(s + j) * 3 + d];
                poly[(size_t)s * 12 + r * 3 + d] = 0.5 * v;
            }
    for (int d = 0; d < 3; d++) poly[(size_t)t->n_seg * 12 + d] = cp[(size_t)(n_points - 1) * 3 + d];
    // root coefficient rows of the primitive (already scaled by translation_maxima), then the root quaternion of the first control point
    const int NB = p->NB, D = p->D, L = p->L;
    t->rows = NB * 3 + 4;
    std::vector<double> E((size_t)t->rows * L, 0.0), mean(t->rows, 0.0);
    for (int i = 0; i < NB; i++)
        for (int d = 0; d < 3; d++) {
            mean[(size_t)i * 3 + d] = p->means_[(size_t)i * D + d];
            for (int k = 0; k < L; k++) E[((size_t)i * 3 + d) * L + k] = p->Es[((size_t)i * D + d) * L + k];
        }
    for (int e = 0; e < 4; e++) {
        const size_t row = (size_t)NB * 3 + e;
        if (D >= 7) {
            mean[row] = p->means_[3 + e];
            for (int k = 0; k < L; k++) E[row * L + k] = p->Es[(size_t)(3 + e) * L + k];
        } else {
            mean[row] = e == 0 ? 1.0 : 0.0;
        }
    }
    // the arc-length parameterisation (splines/arc_length_map.py:45-71): the polyline over granularity + 1 samples, accumulated and
    // normalised -- what the constraints that look targets up BY ARC LENGTH search (mg_frame_constraints.hip)
    std::vector<double> arc((size_t)granularity + 1, 0.0);
    {
        auto point = [&](double u, double *q) {
            const double scaled = t->n_seg * u;
            int index = (int)std::floor(scaled);
            if (index >= t->n_seg) { for (int d = 0; d < 3; d++) q[d] = poly[(size_t)t->n_seg * 12 + d]; return; }
            const double tt = scaled - index;
            const double *A = &poly[(size_t)index * 12];
            for (int d = 0; d < 3; d++) q[d] = ((A[d] * tt + A[3 + d]) * tt + A[6 + d]) * tt + A[9 + d];
        };
        double last[3], cur[3], acc = 0.0;
        point(0.0, last);
        for (int k = 1; k <= granularity; k++) {
            point(k / (double)granularity, cur);
            acc += std::sqrt((cur[0] - last[0]) * (cur[0] - last[0]) + (cur[1] - last[1]) * (cur[1] - last[1]) + (cur[2] - last[2]) * (cur[2] - last[2]));
            arc[(size_t)k] = acc;
            for (int d = 0; d < 3; d++) last[d] = cur[d];
        }
        t->full_arc = acc;
        if (acc > 0.0) for (double &v : arc) v /= acc;
    }
    int rc = mg_traj_upload(poly, &t->d_poly);
    if (rc == MG_OK) rc = mg_traj_upload(arc, &t->d_arc);
    if (rc == MG_OK) rc = mg_traj_upload(E, &t->d_E);
    if (rc == MG_OK) rc = mg_traj_upload(mean, &t->d_mean);
    if (rc != MG_OK) { mg_trajectory_destroy(t); return rc; }
    *out = t;
    return MG_OK;
}

// arguments of one scorer, validated (who: the entry point's name for the message)
static int mg_traj_fill_args(const char *who, mg_primitive *p, const mg_trajectory *t, const mg_time_grid *g, const void *lat, int dt, int64_t B,
                             int64_t ld, double min_u, double weight, const mg_alignment_desc *al, double *errors_dev,
                             int accumulate, double *residuals_dev, mg_traj_args *out) {
    if (!p || !t || t->prim != p || B < 0 || (dt != MG_F32 && dt != MG_F64) || ld < p->L || !(min_u >= 0.0 && min_u <= 1.0) ||
        !std::isfinite(weight)) {
        mg_set_error("%s: bad arguments (trajectory of another primitive, ld < n_components, min_u outside [0, 1] ...)", who);
        return MG_ERR_INVALID_ARGUMENT;
    }
    if (!g) g = p->canonical;
    if (g->prim != p) { mg_set_error("%s: grid belongs to another primitive", who); return MG_ERR_INVALID_ARGUMENT; }
    if (B > 0 && (!lat || !errors_dev)) { mg_set_error("%s: NULL pointer", who); return MG_ERR_INVALID_ARGUMENT; }
    mg_traj_args a;
    a.poly = t->d_poly; a.E = t->d_E; a.mean = t->d_mean; a.lat = lat; a.i0 = g->d_i0; a.w = g->d_w; a.out = errors_dev; a.res = residuals_dev;
    a.B = B; a.ld = ld; a.T = g->T; a.L = p->L; a.NB = p->NB; a.n_seg = t->n_seg; a.G = t->granularity; a.lat_f64 = dt == MG_F64 ? 1 : 0;
    a.accumulate = accumulate ? 1 : 0; a.min_u = min_u; a.weight = weight; a.points = nullptr; a.res_u = nullptr; a.res_n = nullptr; a.paths = nullptr;
    a.search = p->ctx->opt[MG_OPT_TRAJECTORY_SEARCH] == 1 ? 1 : 0;
    a.align_mode = 0;
    for (double &v : a.al) v = 0.0;
    if (al) {
        const double hn = std::sqrt(al->heading[0] * al->heading[0] + al->heading[1] * al->heading[1]);
        if (!(hn > 0.0) || !std::isfinite(hn)) { mg_set_error("%s: heading is zero or not finite", who); return MG_ERR_INVALID_ARGUMENT; }
        if (al->joint != 0 && al->joint != MG_ALIGN_START_POSE) {
            mg_set_error("%s: only the root joint or a start pose can be the aligning reference here (joint %d)", who, al->joint);
            return MG_ERR_UNSUPPORTED;
        }
        a.align_mode = al->joint == MG_ALIGN_START_POSE ? 2 : 1;
        a.al[0] = al->heading[0] / hn; a.al[1] = al->heading[1] / hn; a.al[2] = al->position[0]; a.al[3] = al->position[2];
        if (a.align_mode == 2) a.al[4] = al->position[1];
        else { a.al[4] = al->ref_dir[0]; a.al[5] = al->ref_dir[1]; a.al[6] = al->ref_dir[2]; }
    }
    *out = a;
    return MG_OK;
}

extern "C" int mg_score_trajectory(mg_primitive *p, const mg_trajectory *t, const mg_time_grid *g, const void *lat, int dt, int64_t B,
                                   int64_t ld, double min_u, double weight, const mg_alignment_desc *al, double *errors_dev,
                                   int accumulate, double *residuals_dev) {
    mg_traj_args a;
    { const int rc = mg_traj_fill_args("mg_score_trajectory", p, t, g, lat, dt, B, ld, min_u, weight, al, errors_dev, accumulate, residuals_dev, &a); if (rc != MG_OK) return rc; }
    if (B == 0) return MG_OK;
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    size_t lds = (size_t)(p->L + t->rows) * MG_TRAJ_BLOCK * 8;
    if (lds > 150 * 1024) { mg_set_error("mg_score_trajectory: %d basis functions x %d components do not fit LDS", p->NB, p->L); return MG_ERR_UNSUPPORTED; }
    const size_t poly_bytes = ((size_t)t->n_seg * 12 + 3) * 8;
    const bool poly_lds = lds + poly_bytes <= 158 * 1024;
    if (poly_lds) lds += poly_bytes;
    // (a property of kernel AND device: once per context, not per process -- ADVICE r3)
    if (!(p->ctx->attr_traj & 1u)) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_trajectory_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_trajectory_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        p->ctx->attr_traj |= 1u;
    }
    const int grid = (int)((B + MG_TRAJ_BLOCK - 1) / MG_TRAJ_BLOCK);
    int lanes = mg_traj_lanes(p->ctx, B, B);
    const int coop_cands = MG_TRAJ_COOP_BLOCK / std::max(lanes, 1);
    const size_t coop_lds = (size_t)coop_cands * (p->L + t->rows) * 8 + poly_bytes;
    if (coop_lds > 60 * 1024) lanes = 1;
    const size_t stream_lds = (size_t)p->L * MG_TRAJ_BLOCK * 8 + poly_bytes;
    const dim3 coop_grid((unsigned)((B + coop_cands - 1) / coop_cands));
    if (lanes == 1 && stream_lds <= 60 * 1024 && a.search == 0) { const int rcp = mg_traj_paths(p->ctx, (size_t)B * a.T * 24, &a.paths); if (rcp != MG_OK) return rcp; }
    mg_prof_begin(p->ctx, 10);
    if (lanes == 8) hipLaunchKernelGGL(mg_trajectory_coop_kernel<8>, coop_grid, dim3(MG_TRAJ_COOP_BLOCK), coop_lds, p->ctx->stream, a);
    else if (lanes == 4) hipLaunchKernelGGL(mg_trajectory_coop_kernel<4>, coop_grid, dim3(MG_TRAJ_COOP_BLOCK), coop_lds, p->ctx->stream, a);
    else if (stream_lds <= 60 * 1024) hipLaunchKernelGGL(mg_trajectory_stream_kernel, dim3(grid), dim3(MG_TRAJ_BLOCK), stream_lds, p->ctx->stream, a);
    else if (poly_lds) hipLaunchKernelGGL(mg_trajectory_kernel<true>, dim3(grid), dim3(MG_TRAJ_BLOCK), lds, p->ctx->stream, a);
    else hipLaunchKernelGGL(mg_trajectory_kernel<false>, dim3(grid), dim3(MG_TRAJ_BLOCK), lds, p->ctx->stream, a);
    mg_prof_end(p->ctx, 10);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// n scorers of the same batch size side by side in one launch (canonical grids, no residual vectors): option k's candidates
// lat_dev[k] (B, ld[k]) against trajectories[k], errors_dev[k] written or added to.  alignments: NULL, or n records of which any may be
// NULL.  Where the eight-lane walk does not apply (more than 65 536 candidates, MG_OPT_TRAJECTORY_LANES 1, rows that do not fit) the
// scorers are launched one after the other -- the same results either way.
extern "C" int mg_score_trajectories(int32_t n, mg_primitive *const *prims, const mg_trajectory *const *trajectories, const void *const *lat_dev, int dt,
                                     int64_t B, const int64_t *ld, const double *min_u, const double *weight, const mg_alignment_desc *const *alignments,
                                     double *const *errors_dev, int accumulate) {
    if (n < 0 || (n > 0 && (!prims || !trajectories || !lat_dev || !ld || !min_u || !weight || !errors_dev))) {
        mg_set_error("mg_score_trajectories: bad arguments");
        return MG_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return MG_OK;
    for (int k = 0; k < n; k++)
        if (!prims[k] || prims[k]->ctx != prims[0]->ctx) { mg_set_error("mg_score_trajectories: the primitives must share one context"); return MG_ERR_INVALID_ARGUMENT; }
    mg_context *ctx = prims[0]->ctx;
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    const int lanes = mg_traj_lanes(ctx, B, (int64_t)n * B);
    const bool coop = lanes > 1;
    const int coop_cands = MG_TRAJ_COOP_BLOCK / lanes;
    bool together = B > 0 && n > 1;
    size_t lds = 0;
    for (int k = 0; k < n && together; k++) {
        if (!trajectories[k]) { mg_set_error("mg_score_trajectories: trajectory %d is NULL", k); return MG_ERR_INVALID_ARGUMENT; }
        const size_t poly_bytes = ((size_t)trajectories[k]->n_seg * 12 + 3) * 8;
        const size_t need = coop ? (size_t)coop_cands * (prims[k]->L + trajectories[k]->rows) * 8 + poly_bytes
                                 : (size_t)prims[k]->L * MG_TRAJ_BLOCK * 8 + poly_bytes;
        if (need > 60 * 1024) together = false;
        lds = std::max(lds, need);
    }
    if (!together) {
        for (int k = 0; k < n; k++) {
            const int rc = mg_score_trajectory(prims[k], trajectories[k], nullptr, lat_dev[k], dt, B, ld[k], min_u[k], weight[k], alignments ? alignments[k] : nullptr,
                                               errors_dev[k], accumulate, nullptr);
            if (rc != MG_OK) return rc;
        }
        return MG_OK;
    }
    const int cands_per_wg = coop ? coop_cands : MG_TRAJ_BLOCK;
    const int per = (int)((B + cands_per_wg - 1) / cands_per_wg);
    for (int k0 = 0; k0 < n; k0 += MG_TRAJ_MULTI_MAX) {
        mg_traj_multi m;
        memset(&m, 0, sizeof(m));
        m.n = std::min(MG_TRAJ_MULTI_MAX, n - k0);
        for (int i = 0; i < m.n; i++) {
            const int k = k0 + i;
            const int rc = mg_traj_fill_args("mg_score_trajectories", prims[k], trajectories[k], nullptr, lat_dev[k], dt, B, ld[k], min_u[k], weight[k],
                                             alignments ? alignments[k] : nullptr, errors_dev[k], accumulate, nullptr, &m.a[i]);
            if (rc != MG_OK) return rc;
            m.wg0[i + 1] = m.wg0[i] + per;
        }
        if (!coop && m.a[0].search == 0) {      // the streaming kernels' root paths: one block of the context's buffer per scorer
            size_t off = 0;
            double *base = nullptr;
            for (int i = 0; i < m.n; i++) off += (size_t)B * m.a[i].T * 3;
            const int rcp = mg_traj_paths(ctx, off * 8, &base);
            if (rcp != MG_OK) return rcp;
            off = 0;
            for (int i = 0; i < m.n; i++) { m.a[i].paths = base + off; off += (size_t)B * m.a[i].T * 3; }
        }
        mg_prof_begin(ctx, 10);
        if (lanes == 8) hipLaunchKernelGGL(mg_trajectory_coop_multi_kernel<8>, dim3((unsigned)m.wg0[m.n]), dim3(MG_TRAJ_COOP_BLOCK), lds, ctx->stream, m);
        else if (lanes == 4) hipLaunchKernelGGL(mg_trajectory_coop_multi_kernel<4>, dim3((unsigned)m.wg0[m.n]), dim3(MG_TRAJ_COOP_BLOCK), lds, ctx->stream, m);
        else hipLaunchKernelGGL(mg_trajectory_stream_multi_kernel, dim3((unsigned)m.wg0[m.n]), dim3(MG_TRAJ_BLOCK), lds, ctx->stream, m);
        mg_prof_end(ctx, 10);
        MG_HIP_CHECK(hipGetLastError());
    }
    return MG_OK;
}

// The same search for positions the caller supplies: points_dev (B, T, 3) float64 (e.g. one joint's track from mg_joint_positions,
// aligned by the caller) -- TrajectoryConstraint for joints other than the root (trajectory_constraint.py:95-121 with
// skeleton.nodes[joint].get_global_position(frame)).
static int mg_traj_points_launch(const char *who, mg_primitive *p, const mg_trajectory *t, const double *points_dev, int64_t B, int32_t T, double min_u,
                                 double weight, double *errors_dev, int accumulate, double *residuals_dev, double *params_dev, int32_t *evals_dev = nullptr) {
    if (!p || !t || t->prim != p || B < 0 || T < 1 || !(min_u >= 0.0 && min_u <= 1.0) || !std::isfinite(weight)) {
        mg_set_error("%s: bad arguments", who);
        return MG_ERR_INVALID_ARGUMENT;
    }
    if (B == 0) return MG_OK;
    if (!points_dev || (!errors_dev && !params_dev && !residuals_dev && !evals_dev)) { mg_set_error("%s: NULL pointer", who); return MG_ERR_INVALID_ARGUMENT; }
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    mg_traj_args a;
    a.poly = t->d_poly; a.E = t->d_E; a.mean = t->d_mean; a.lat = nullptr; a.i0 = nullptr; a.w = nullptr; a.out = errors_dev; a.res = residuals_dev;
    a.B = B; a.ld = 0; a.T = T; a.L = p->L; a.NB = p->NB; a.n_seg = t->n_seg; a.G = t->granularity; a.lat_f64 = 1;
    a.accumulate = accumulate ? 1 : 0; a.min_u = min_u; a.weight = weight; a.points = points_dev; a.res_u = params_dev; a.res_n = evals_dev; a.paths = nullptr;
    a.search = p->ctx->opt[MG_OPT_TRAJECTORY_SEARCH] == 1 ? 1 : 0;
    a.align_mode = 0;
    for (double &v : a.al) v = 0.0;
    const int grid = (int)((B + MG_TRAJ_BLOCK - 1) / MG_TRAJ_BLOCK);
    const size_t poly_bytes = ((size_t)t->n_seg * 12 + 3) * 8;
    if (!(p->ctx->attr_traj & 1u)) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_trajectory_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_trajectory_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        p->ctx->attr_traj |= 1u;
    }
    mg_prof_begin(p->ctx, 10);
    const int lanes = poly_bytes <= 60 * 1024 ? mg_traj_lanes(p->ctx, B, B) : 1;
    if (lanes == 8) hipLaunchKernelGGL(mg_trajectory_coop_kernel<8>, dim3((unsigned)((B + 7) / 8)), dim3(MG_TRAJ_COOP_BLOCK), poly_bytes, p->ctx->stream, a);
    else if (lanes == 4) hipLaunchKernelGGL(mg_trajectory_coop_kernel<4>, dim3((unsigned)((B + 15) / 16)), dim3(MG_TRAJ_COOP_BLOCK), poly_bytes, p->ctx->stream, a);
    else if (poly_bytes <= 158 * 1024) hipLaunchKernelGGL(mg_trajectory_kernel<true>, dim3(grid), dim3(MG_TRAJ_BLOCK), poly_bytes, p->ctx->stream, a);
    else hipLaunchKernelGGL(mg_trajectory_kernel<false>, dim3(grid), dim3(MG_TRAJ_BLOCK), 0, p->ctx->stream, a);
    mg_prof_end(p->ctx, 10);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}
extern "C" int mg_score_trajectory_points(mg_primitive *p, const mg_trajectory *t, const double *points_dev, int64_t B, int32_t T, double min_u,
                                          double weight, double *errors_dev, int accumulate, double *residuals_dev) {
    if (B > 0 && !errors_dev) { mg_set_error("mg_score_trajectory_points: NULL pointer"); return MG_ERR_INVALID_ARGUMENT; }
    return mg_traj_points_launch("mg_score_trajectory_points", p, t, points_dev, B, T, min_u, weight, errors_dev, accumulate, residuals_dev, nullptr);
}
// ParameterizedSpline.find_closest_point_fast (splines/parameterized_spline.py:303-322) for a batch of point sequences, chained the way
// TrajectoryConstraint.get_residual_vector chains it (trajectory_constraint.py:103-113: every frame's search is bounded below by, and
// started at, the previous frame's parameter): params_dev (B, T) the parameter of every frame's point, distances_dev (B, T) its
// distance, evaluations_dev (B, T) int32 the (f, g) evaluations the frame's search took -- scipy's nfev / 2; the reference's search
// only -- (any may be NULL).
extern "C" int mg_trajectory_closest_points(mg_primitive *p, const mg_trajectory *t, const double *points_dev, int64_t B, int32_t T, double min_u,
                                            double *params_dev, double *distances_dev, int32_t *evaluations_dev) {
    if (B > 0 && !params_dev && !distances_dev && !evaluations_dev) { mg_set_error("mg_trajectory_closest_points: nothing to write"); return MG_ERR_INVALID_ARGUMENT; }
    return mg_traj_points_launch("mg_trajectory_closest_points", p, t, points_dev, B, T, min_u, 1.0, nullptr, 0, distances_dev, params_dev, evaluations_dev);
}

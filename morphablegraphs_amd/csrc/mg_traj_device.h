// Device-side pieces of the trajectory constraints (gfx950): the Catmull-Rom target spline and the monotone closest-point search,
// shared by mg_trajectory.hip (the root's own path, or any joint's track) and mg_frame_constraints.hip (the per-frame
// constraints scored as a list) so that both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

__device__ __forceinline__ void mg_traj_point(const double *__restrict__ poly, int n_seg, double u, double *p) {
    const double scaled = n_seg * u;
    int index = (int)floor(scaled);
    index = index < n_seg ? index : n_seg;
    if (index >= n_seg) {                       // past the last segment: the last control point
        const double *q = poly + (size_t)n_seg * 12;
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
        return;
    }
    const double t = scaled - index;
    const double *A = poly + (size_t)index * 12;
#pragma unroll
    for (int d = 0; d < 3; d++) p[d] = ((A[d] * t + A[3 + d]) * t + A[6 + d]) * t + A[9 + d];
}
__device__ __forceinline__ double mg_traj_d2(const double *poly, int n_seg, double u, const double *q) {
    double p[3];
    mg_traj_point(poly, n_seg, u, p);
    const double x = p[0] - q[0], y = p[1] - q[1], z = p[2] - q[2];
    return x * x + y * y + z * z;
}

// squared distance, and its first and second derivative in u (the segment's cubic differentiated; false past the last segment)
__device__ __forceinline__ bool mg_traj_d2_derivs(const double *poly, int n_seg, double u, const double *q, double *f0, double *f1, double *f2) {
    const double scaled = n_seg * u;
    int index = (int)floor(scaled);
    if (index >= n_seg) return false;
    const double t = scaled - index;
    const double *A = poly + (size_t)index * 12;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double v = ((A[d] * t + A[3 + d]) * t + A[6 + d]) * t + A[9 + d] - q[d];
        const double p1 = ((3.0 * A[d] * t + 2.0 * A[3 + d]) * t + A[6 + d]) * n_seg;
        const double p2 = (6.0 * A[d] * t + 2.0 * A[3 + d]) * n_seg * n_seg;
        s0 += v * v; s1 += v * p1; s2 += p1 * p1 + v * p2;
    }
    *f0 = s0; *f1 = 2.0 * s1; *f2 = 2.0 * s2;
    return true;
}

// Up to four Newton steps on the squared distance inside the bracket [lo, hi] of the grid minimum (one-sided at the ends of the
// range), from *u_io clamped into it: the bracket's local minimum to rounding; a step that does not lower the distance ends the
// refinement.  Returns the squared distance at the refined *u_io.  The search's statement evaluates the distance at a trial point and,
// when the step is taken, the distance and its derivatives there again: mg_traj_d2_derivs' f0 IS mg_traj_d2's value, operation for
// operation ((x^2 + y^2) + z^2 of the same Horner forms; no contraction), so one evaluation per step serves both -- the walk is a chain
// of dependent evaluations, and this halves its Newton part.
__device__ __forceinline__ double mg_traj_refine(const double *poly, int n_seg, double lo, double hi, double *u_io, const double *q) {
    double u = fmin(hi, fmax(lo, *u_io));
    double f0, f1, f2;
    bool ok = mg_traj_d2_derivs(poly, n_seg, u, q, &f0, &f1, &f2);
    if (!ok) f0 = mg_traj_d2(poly, n_seg, u, q);           // past the last segment: the last control point's distance, no derivatives
    for (int it = 0; it < 4; it++) {
        if (!ok || !(f2 > 0.0)) break;
        const double un = fmin(hi, fmax(lo, u - f1 / f2));
        if (un == u) break;                                 // a fixed point: every further step would repeat this one
        double g0, g1, g2;
        const bool okn = mg_traj_d2_derivs(poly, n_seg, un, q, &g0, &g1, &g2);
        if (!okn) g0 = mg_traj_d2(poly, n_seg, un, q);
        if (g0 > f0) break;
        u = un; ok = okn; f0 = g0; f1 = g1; f2 = g2;
    }
    *u_io = u;
    return f0;
}

// The distance from q to the closest point of the spline whose parameter is at or after *min_u, and that parameter back in *min_u
// (the bound of the next frame's search): on the grid u_k = k / G walk forward from the bound while the squared distance falls,
// refine by the parabola through the three values around the minimum, then by up to four Newton steps inside that bracket
// (oracle/mg_oracle.py closest_point_walk; trajectory_constraint.py:113-131 bounds the search the same way).
// WAVE_HELP (only where all 64 lanes of the wave reach the call together): a lane whose walk has not ended after MG_TRAJ_WALK_ALONE
// steps gets the whole wave -- the next 64 grid values side by side for ITS point, the first that does not fall found by a ballot --
// lane by lane.  With one lane per candidate a wave walks as long as its longest lane: candidates far from the spline (hundreds of
// grid steps in one frame) cost every other lane of the wave that time; helped, such a walk is a few rounds.  The same grid values
// in the same comparisons: the same k, the same bits.
#define MG_TRAJ_WALK_ALONE 16
template <bool WAVE_HELP = false>
__device__ __forceinline__ double mg_traj_closest_dist(const double *__restrict__ poly, int n_seg, int G, double invG, double *min_u_io, const double *q) {
    const double min_u = *min_u_io;
    struct { const double *poly; int n_seg; } a = {poly, n_seg};
    // closest point at or after min_u: grid walk + parabola (oracle closest_point_walk)
    int k = (int)ceil(min_u * G - 1e-12);
    k = k < G ? k : G;
    double dk = mg_traj_d2(a.poly, a.n_seg, k * invG, q);
    const double d_start = mg_traj_d2(a.poly, a.n_seg, min_u, q);
    if constexpr (!WAVE_HELP) {
        while (k < G) {
            const double dn = mg_traj_d2(a.poly, a.n_seg, (k + 1) * invG, q);
            if (dn >= dk) break;
            k++;
            dk = dn;
        }
    } else {
        bool done = false;
        for (int step = 0; step < MG_TRAJ_WALK_ALONE && !done; step++) {
            if (k >= G) { done = true; break; }
            const double dn = mg_traj_d2(a.poly, a.n_seg, (k + 1) * invG, q);
            if (dn >= dk) done = true;
            else { k++; dk = dn; }
        }
        const int lane = (int)__lane_id();
        unsigned long long pend = __ballot(!done);
        while (pend) {                                   // (uniform: every lane runs every helped lane's rounds)
            const int src = __ffsll((long long)pend) - 1;
            const double qs[3] = {__shfl(q[0], src), __shfl(q[1], src), __shfl(q[2], src)};
            int ks = __shfl(k, src);
            double dks = __shfl(dk, src);
            for (;;) {
                const int idx = ks + 1 + lane;           // lane l: would the walk, standing at idx - 1, stop there?
                const double val = mg_traj_d2(a.poly, a.n_seg, (idx < G ? idx : G) * invG, qs);
                const double up = __shfl_up(val, 1);
                const double before = lane == 0 ? dks : up;
                const unsigned long long stop = __ballot(idx > G || val >= before);
                if (stop) {
                    const int first = __ffsll((long long)stop) - 1;
                    const double dfirst = __shfl(val, first > 0 ? first - 1 : 0);
                    if (lane == src) { k = ks + first; dk = first > 0 ? dfirst : dks; }
                    break;
                }
                ks += 64;
                dks = __shfl(val, 63);
            }
            pend &= pend - 1;
        }
    }
    double u = k * invG;
    if (k > 0 && k < G) {
        const double da = mg_traj_d2(a.poly, a.n_seg, (k - 1) * invG, q), dc = mg_traj_d2(a.poly, a.n_seg, (k + 1) * invG, q);
        const double den = da - 2.0 * dk + dc;
        if (den > 0.0) u = (k + 0.5 * (da - dc) / den) * invG;
    }
    double d2 = mg_traj_refine(a.poly, a.n_seg, fmax(min_u, (k - 1) * invG), fmin(1.0, (k + 1) * invG), &u, q);
    const double uc = fmin(1.0, fmax(min_u, u));            // (the bracket lies inside [min_u, 1]: a no-op kept from the statement of the search)
    if (uc != u) { u = uc; d2 = mg_traj_d2(a.poly, a.n_seg, u, q); }
    if (d_start <= d2) { u = min_u; d2 = d_start; }
    *min_u_io = u;
    return sqrt(d2);
}

// The same search by W consecutive lanes that hold the same q and bound (W a power of two, the group aligned in the wave); every lane
// returns what mg_traj_closest_dist returns, bit for bit: the grid walk -- a chain of dependent evaluations, one per grid step, 6 per
// frame for a path that moves 1/156 of the spline per frame at granularity 1000 -- becomes ONE evaluation per lane (the grid values
// around the bound side by side, the first one that does not fall found by a ballot), the start value rides in the group's last
// lane, the parabola's neighbours are already there.  The Newton steps (a chain by nature) are done by every lane alike.
// For batches that do not fill the chip with one lane per candidate (the walk is latency, not throughput).
template <int W>
__device__ __forceinline__ double mg_traj_closest_dist_coop(const double *__restrict__ poly, int n_seg, int G, double invG, double *min_u_io, const double *q) {
    const int lane = (int)__lane_id();
    const int sub = lane & (W - 1), g0 = lane & ~(W - 1);
    const double min_u = *min_u_io;
    int k = (int)ceil(min_u * G - 1e-12);
    k = k < G ? k : G;
    // first window: lanes 0 .. W-2 the grid values k-1 .. k+W-3, lane W-1 the value at the bound itself
    int base = k - 1, last = W - 2;
    double mine;
    {
        int idx = base + sub;
        idx = idx < 0 ? 0 : (idx > G ? G : idx);
        mine = mg_traj_d2(poly, n_seg, sub == W - 1 ? min_u : idx * invG, q);
    }
    const double d_start = __shfl(mine, g0 + W - 1);
    for (;;) {
        const int idx = base + sub;
        const double next = __shfl_down(mine, 1);
        const bool stop = sub <= last && (idx >= G || (sub < last && next >= mine));
        const unsigned group = (unsigned)(__ballot(stop) >> g0) & ((1u << W) - 1u) & ~((1u << (k - base)) - 1u);
        if (group) { k = base + (__ffs(group) - 1); break; }
        // still falling at the window's end: the next window starts one before it (the parabola wants that neighbour)
        k = base + last;
        base = k - 1; last = W - 1;
        int i2 = base + sub;
        i2 = i2 > G ? G : i2;
        mine = mg_traj_d2(poly, n_seg, i2 * invG, q);
    }
    const double dk = __shfl(mine, g0 + (k - base));
    double u = k * invG;
    if (k > 0 && k < G) {
        const double da = __shfl(mine, g0 + (k - base) - 1), dc = __shfl(mine, g0 + (k - base) + 1);
        const double den = da - 2.0 * dk + dc;
        if (den > 0.0) u = (k + 0.5 * (da - dc) / den) * invG;
    }
    double d2 = mg_traj_refine(poly, n_seg, fmax(min_u, (k - 1) * invG), fmin(1.0, (k + 1) * invG), &u, q);
    const double uc = fmin(1.0, fmax(min_u, u));
    if (uc != u) { u = uc; d2 = mg_traj_d2(poly, n_seg, u, q); }
    if (d_start <= d2) { u = min_u; d2 = d_start; }
    *min_u_io = u;
    return sqrt(d2);
}

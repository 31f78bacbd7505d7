// Device-side pieces of the trajectory constraints (gfx950): the Catmull-Rom target spline and the monotone closest-point search,
// shared by mg_trajectory.hip (the root's own path, or any joint's track) and mg_frame_constraints.hip (the per-frame
// constraints scored as a list) so that both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

// (the reference search and what it evaluates also compile for the host: tests/lbfgsb_host_check.cpp runs the device's very
// statements on the golden tracks without a GPU -- test infrastructure, never linked into the library)
#define MG_HD __host__ __device__

MG_HD __forceinline__ void mg_traj_point(const double *__restrict__ poly, int n_seg, double u, double *p) {
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
MG_HD __forceinline__ double mg_traj_d2(const double *poly, int n_seg, double u, const double *q) {
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

// ---------------------------------------------------------------------------------------------------------------------------------
// THE REFERENCE'S SEARCH (round 5).  ParameterizedSpline.find_closest_point_fast (splines/parameterized_spline.py:303-322) is
// scipy.optimize.minimize(distance, [min_u], method="L-BFGS-B", bounds=[(min_u, 1)]) with a forward-difference gradient.  That is not
// a local search: its first step goes to the far bound and the More'-Thuente line search interpolates back from there, so which
// local minimum of the distance a frame lands in -- and with it the bound of every later frame -- is decided by that algorithm's own
// arithmetic.  Measured against vectors the reference's function produced (tests/golden/trajectory_closest_point.npz), the monotone
// walk above ends up to 0.9 of the parameter range away on 12 of 26 tracks; L-BFGS-B 3.0 restated for ONE bounded variable
// reproduces all 26 chains to 2e-7 of the parameter (oracle/mg_oracle.py lbfgsb_1d, which documents the restatement routine by
// routine and is held to scipy itself).  What follows is that restatement on the device:
//   * objective f(u) = |P(u) - q| (the reference's dist_objective: a norm, not its square), gradient (f(u + h) - f(u)) / h with
//     h = 1e-8, turned around where u + h would leave [lb, 1] (scipy _numdiff.approx_derivative, 2-point, abs_step, bounds);
//   * mainlb for n = 1: the limited-memory matrix is the scalar theta = y'y / s'y (any BFGS update in one dimension gives B = y / s),
//     the generalised Cauchy point is clamp(u - g / theta), a free Cauchy point is the model's minimiser already (subsm: rounding only);
//   * formk's verdict: the routine that prepares subsm fails ("nonpositive definiteness ... refresh the lbfgs memory and restart
//     the iteration") when its incrementally kept matrix is stale, which for one variable is: the variable ENTERED the free set in
//     this iteration and at least two pairs are stored (the iteration before had its Cauchy point on a bound, skipped formk, and
//     the row of the pair before it was never written).  The oracle transcribes formk's bookkeeping and both Cholesky steps in full;
//     the rule agreed with it in all of 108 000 calls (1 900 failures) of a randomised campaign and on every golden track.  In
//     exact arithmetic the stale matrix could still factorise when the newest curvature estimate is below about half the oldest
//     stored one: not observed, and a restart only changes the path to the same local minimum;
//   * lnsrlb + dcsrch + dcstep (ftol 1e-3, gtol 0.9, xtol 0.1, at most 20 evaluations), the two convergence tests (projected gradient
//     <= 1e-5, relative reduction <= 1e7 eps), the curvature test of the update (s'y > eps |g'd| stp).
// Attainable agreement: the forward difference amplifies the last bit of f by 1e8, so two correct implementations agree to ~1e-8 in
// u per search and -- rarely, 0.2 % of single searches against scipy -- stop one iteration apart (|du| < 1e-5).
// One lane per candidate; ~7 spline evaluations per frame on average (2 where the bound holds the point, up to ~70).
MG_HD __forceinline__ double mg_traj_dist(const double *poly, int n_seg, double u, const double *q) {
    return sqrt(mg_traj_d2(poly, n_seg, u, q));
}
// f and the forward-difference gradient at x (scipy: fun_and_grad)
MG_HD __forceinline__ void mg_lb_fg(const double *poly, int n_seg, const double *q, double lb, double x, double *f, double *g) {
    const double f0 = mg_traj_dist(poly, n_seg, x, q);
    double h = 1.0e-8;
    const double lower = x - lb, upper = 1.0 - x;
    if (x + h < lb || x + h > 1.0) {
        if (fabs(h) <= fmax(lower, upper)) h = -h;
        else if (upper >= lower) h = upper;
        else h = -lower;
    }
    const double x1 = x + h;
    const double f1 = mg_traj_dist(poly, n_seg, x1, q);
    *f = f0;
    *g = (f1 - f0) / (x1 - x);
}
// MINPACK-2 dcstep: the safeguarded cubic / quadratic step and the update of the interval of uncertainty.  The routine's four cases
// (higher value; lower value and derivatives of opposite sign; lower value, same sign, smaller derivative; lower value, same sign,
// derivative not smaller) share one shape -- theta, s, gamma = s sqrt(.), r = p / q, a cubic and a quadratic candidate -- with
// different operands: written as ONE path over selected operands (every value the statements of the case at hand would compute, bit for
// bit: a quotient of two negated operands is the same quotient), a wave whose lanes sit in different cases runs it once instead of
// four times (8 divisions and a root instead of up to 28 and 4).
MG_HD __forceinline__ void mg_lb_dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp, const double fp, const double dp,
                                        bool &brackt, const double stpmin, const double stpmax) {
    const double sgnd = dp * (dx / fabs(dx));
    const int cs = fp > fx ? 1 : (sgnd < 0.0 ? 2 : (fabs(dp) < fabs(dx) ? 3 : 4));
    const bool c4 = cs == 4;
    const double stA = c4 ? sty : stx, fA = c4 ? fy : fx, dA = c4 ? dy : dx;     // (case 4 interpolates between stp and sty)
    const double theta = 3.0 * (fA - fp) / (stp - stA) + dA + dp;
    const double s = fmax(fabs(theta), fmax(fabs(dA), fabs(dp)));
    double arg = (theta / s) * (theta / s) - (dA / s) * (dp / s);
    if (cs == 3) arg = fmax(0.0, arg);
    double gamma = s * sqrt(arg);
    if (cs == 1 ? stp < stx : stp > stA) gamma = -gamma;
    const double u = cs == 1 ? dx : dp;
    const double p = (gamma - u) + theta;
    const double q = cs == 3 ? (gamma + (dx - dp)) + gamma : ((gamma - u) + gamma) + (cs == 1 ? dp : dA);
    const double r = p / q;
    double stpc = cs == 1 ? stx + r * (stp - stx) : stp + r * (stA - stp);
    if (cs == 3 && !(r < 0.0 && gamma != 0.0)) stpc = stp > stx ? stpmax : stpmin;
    const double q1 = (fx - fp) / (stp - stx);
    const double stpq = cs == 1 ? stx + ((dx / (q1 + dx)) / 2.0) * (stp - stx) : stp + (dp / (dp - dx)) * (stx - stp);
    double stpf;
    if (cs == 1) stpf = fabs(stpc - stx) < fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
    else if (cs == 2) stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
    else if (cs == 3) {
        if (brackt) {
            stpf = fabs(stpc - stp) < fabs(stpq - stp) ? stpc : stpq;
            stpf = stp > stx ? fmin(stp + 0.66 * (sty - stp), stpf) : fmax(stp + 0.66 * (sty - stp), stpf);
        } else {
            stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
            stpf = fmin(stpmax, stpf);
            stpf = fmax(stpmin, stpf);
        }
    } else stpf = brackt ? stpc : (stp > stx ? stpmax : stpmin);
    if (cs <= 2) brackt = true;
    if (fp > fx) { sty = stp; fy = fp; dy = dp; }
    else {
        if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
        stx = stp; fx = fp; dx = dp;
    }
    stp = stpf;
}
// One search of the reference's (oracle/mg_oracle.py closest_point_lbfgsb, statement for statement, with formk's verdict by the rule
// above) as a STATE MACHINE: begin(), then trip() until done -- every trip is one evaluation of (f, g) followed by the bookkeeping
// that decides the next point.  Two things make that form the device's: (i) written as the algorithm's nested loops (iterations around
// line-search trials) a wave runs the SUM over iterations of the longest line search any of its 64 lanes has; flat, the longest lane's
// total; (ii) a lane's frames chain only through its OWN bound, so with the state in a struct the lanes of a wave need not walk the
// frames in step (mg_traj_chain below): a wave then runs its slowest lane's total over all frames (~1.6 x the mean) instead of the
// sum over frames of each frame's slowest lane (8.7 x the mean: the reference's line searches that end at the noise floor after 20
// trials are rare per lane and frame but present in nearly every frame of SOME lane of 64).  Nested loops 12.3 ms, flat per frame 10.5 ms,
// lanes out of step 5.2 ms per 4096 candidates x 156 frames; what is left is the slowest candidate's own chain: ~2500 trips of ~1260
// instructions, issued at 4 cycles each by the one wave its SIMD holds.
struct mg_lb_search {
    double lb, x, f, g;
    double t, r_, fold, d, z, stpmx, gdold, theta;                                                  // the iteration
    double stp, finit, ginit, gtest, width, width1, stx, fx, gx, sty, fy, gy, stmin, stmax;         // the line search
    int col, itr, nit, ifun, stage, trips;
    bool was_free, brackt, searching, done;

    MG_HD __forceinline__ void begin(double min_u) {
        lb = min_u; x = min_u; f = 0.0; g = 0.0;
        t = r_ = fold = d = z = stpmx = gdold = 0.0; theta = 1.0;
        stp = 1.0; finit = ginit = gtest = width = width1 = stx = fx = gx = sty = fy = gy = stmin = stmax = 0.0;
        col = itr = nit = ifun = trips = 0; stage = 1;
        was_free = true; brackt = false; searching = false; done = false;
    }
    MG_HD __forceinline__ double projgr(double xx, double gg) const { return fabs(gg < 0.0 ? fmax(xx - 1.0, gg) : fmin(xx - lb, gg)); }

    MG_HD __forceinline__ void trip(const double *poly, int n_seg, const double *q) {
        const double ub = 1.0;
        const double epsmch = 2.220446049250313e-16, pgtol = 1.0e-5, factr = 1.0e7;
        const double ftol = 1.0e-3, gtol = 0.9, xtol = 0.1, stpmin = 0.0;
        double fn, gn;
        trips++;
        mg_lb_fg(poly, n_seg, q, lb, x, &fn, &gn);
        bool start = false;                 // set up a new iteration (Cauchy point, line search) from (x, f, g)
        if (!searching) {                   // the start point
            f = fn; g = gn;
            if (lb == ub || projgr(x, g) <= pgtol) done = true;     // (lb == ub: minimize() returns the bound without a search)
            else start = true;
        } else {                            // a trial point of the line search: dcsrch's re-entry
            f = fn; g = gn;
            const double gd = g * d;
            const double ftest = finit + stp * gtest;
            if (stage == 1 && f <= ftest && gd >= 0.0) stage = 2;
            bool ended = false;
            if (brackt && (stp <= stmin || stp >= stmax)) ended = true;
            if (brackt && stmax - stmin <= xtol * stmax) ended = true;
            if (stp == stpmx && f <= ftest && gd <= gtest) ended = true;
            if (stp == stpmin && (f > ftest || gd >= gtest)) ended = true;
            if (f <= ftest && fabs(gd) <= gtol * (-ginit)) ended = true;
            if (ended) {                    // NEW_X
                searching = false;
                itr++; nit++;
                if (projgr(x, g) <= pgtol || fold - f <= epsmch * factr * fmax(fabs(fold), fmax(fabs(f), 1.0)) || nit >= 15000) done = true;
                else {
                    const double y = g - r_, rr = y * y;
                    double dr, dd;
                    if (stp == 1.0) { dr = gd - gdold; dd = -gdold; }
                    else { dr = (gd - gdold) * stp; dd = -gdold * stp; }
                    if (!(dr <= epsmch * dd)) { col = col < 10 ? col + 1 : 10; theta = rr / dr; }     // (else: the update is skipped, the model stays)
                    start = true;
                }
            } else {
                if (stage == 1 && f <= fx && f > ftest) {
                    const double fm = f - stp * gtest, gm = gd - gtest;
                    double fxm = fx - stx * gtest, fym = fy - sty * gtest, gxm = gx - gtest, gym = gy - gtest;
                    mg_lb_dcstep(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax);
                    fx = fxm + stx * gtest; fy = fym + sty * gtest; gx = gxm + gtest; gy = gym + gtest;
                } else {
                    mg_lb_dcstep(stx, fx, gx, sty, fy, gy, stp, f, gd, brackt, stmin, stmax);
                }
                if (brackt) {
                    if (fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
                    width1 = width;
                    width = fabs(sty - stx);
                }
                if (brackt) { stmin = fmin(stx, sty); stmax = fmax(stx, sty); }
                else { stmin = stp + 1.1 * (stp - stx); stmax = stp + 4.0 * (stp - stx); }
                stp = fmax(stp, stpmin);
                stp = fmin(stp, stpmx);
                if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax)) stp = stx;
                ifun++;
                if (ifun - 1 >= 20) {       // the line search gave up: back to the iteration's start point
                    x = t; g = r_; f = fold;
                    searching = false;
                    if (col == 0) done = true;                       // abnormal termination in the line search
                    else { col = 0; theta = 1.0; start = true; }
                } else x = stp == 1.0 ? z : stp * d + t;
            }
        }
        while (start) {                     // (at most three passes: a formk restart, a restart after a non-descent direction)
            start = false;
            // cauchy
            const double neggi = -g;
            const double tl = x - lb, tu = ub - x;
            const bool xlower = tl <= 0.0, xupper = tu <= 0.0;
            const bool fixed = (xlower && neggi <= 0.0) || (!xlower && xupper && neggi >= 0.0);
            bool fr;
            if (fixed || neggi == 0.0) { z = x; fr = !fixed; }
            else {
                const double f1 = -neggi * neggi, f2 = -theta * f1;
                double dtm = -f1 / f2;
                const double tbreak = neggi < 0.0 ? tl / (-neggi) : tu / neggi;
                if (dtm < tbreak) { if (dtm <= 0.0) dtm = 0.0; z = x + dtm * neggi; fr = true; }
                else { z = neggi > 0.0 ? ub : lb; fr = false; }
            }
            // freev + formk's verdict
            const bool entered = itr > 0 && fr && !was_free;
            was_free = fr;
            if (fr && entered && col >= 2) { col = 0; theta = 1.0; start = true; continue; }
            d = z - x;
            // lnsrlb
            if (itr == 0) stpmx = 1.0;
            else {
                stpmx = 1.0e10;
                if (d < 0.0) { const double a2 = lb - x; if (a2 >= 0.0) stpmx = 0.0; else if (d * stpmx < a2) stpmx = a2 / d; }
                else if (d > 0.0) { const double a2 = ub - x; if (a2 <= 0.0) stpmx = 0.0; else if (d * stpmx > a2) stpmx = a2 / d; }
            }
            t = x; r_ = g; fold = f;
            const double gd = g * d;
            gdold = gd;
            if (gd >= 0.0) {                // not a descent direction: restart without memory, or give up
                if (col == 0) { done = true; break; }
                col = 0; theta = 1.0; start = true;
                continue;
            }
            // dcsrch, START
            stp = 1.0; brackt = false; stage = 1; ifun = 1;
            finit = f; ginit = gd; gtest = ftol * ginit;
            width = stpmx - stpmin; width1 = width / 0.5;
            stx = 0.0; fx = finit; gx = ginit; sty = 0.0; fy = finit; gy = ginit; stmin = 0.0; stmax = stp + 4.0 * stp;
            searching = true;
            x = z;                          // (stp = 1)
        }
    }
};
// The distance from q to the spline point the reference's search settles on, its parameter back in *min_u_io (the next frame's bound
// and start); n_trips: the (f, g) evaluations the search took (scipy's nfev / 2).
MG_HD __forceinline__ double mg_traj_closest_lbfgsb(const double *poly, int n_seg, double *min_u_io, const double *q, int *n_trips = nullptr) {
    mg_lb_search s;
    s.begin(*min_u_io);
    while (!s.done) s.trip(poly, n_seg, q);
    *min_u_io = s.x;
    if (n_trips) *n_trips = s.trips;
    return s.f;
}
// TrajectoryConstraint.get_residual_vector's chain over a lane's own T positions (trajectory_constraint.py:103-113): getq(f, q) hands
// over frame f's position, emit(f, distance, parameter, trips) takes every frame's result in frame order.  The lanes of a wave are
// NOT held in step: a lane whose search is done moves on to its next frame while its neighbours are still searching.
template <typename GetQ, typename Emit>
MG_HD __forceinline__ void mg_traj_chain(const double *poly, int n_seg, int T, double min_u, GetQ getq, Emit emit) {
    mg_lb_search s;
    double q[3] = {0.0, 0.0, 0.0};
    int f = 0;
    if (T > 0) { getq(0, q); s.begin(min_u); }
    while (f < T) {
        s.trip(poly, n_seg, q);
        if (s.done) {
            emit(f, s.f, s.x, s.trips);
            min_u = s.x;
            f++;
            if (f < T) { getq(f, q); s.begin(min_u); }
        }
    }
}

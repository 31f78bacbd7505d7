// Device-side pieces of the fused constraint scorer (gfx950): the argument block and the weighted residual of one
// constraint for one candidate, shared by the stand-alone scoring kernels (mg_score.hip) and the one-launch planner
// step (mg_options.hip) so that both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "mg_internal.h"

struct mg_score_args {
    const double *W;      // [rows][L]     sum_j w_j E'[(i0+j) D + d]; constraint c owns rows woff[c] ..
    const double *bias;   // [rows]        mean frame at t_c
    const double *par;    // [n][8]        type, weight, target[3], ref_dir[3]
    const int32_t *woff;  // [n + 1]
    const int32_t *chain; // [n]           FK chain length
    const double *choff;  // [n][2][MG_MAX_CHAIN][3]
    const double *pose;   // pose constraints' tables (MG_POSE_HDR / MG_POSE_REC layout) or NULL
    const double *align;  // [8] or NULL: chain length, previous heading (x,z), previous root (x,z), ref_dir; rows at woff[n]
    const double *align_cand;   // NULL, or (B, 4): the previous heading (x, z) and root position (x, z) of EVERY candidate, in
                                // place of align[1..4] (the steps of a graph walk: a candidate's step is aligned to ITS OWN previous step)
    const void *lat;
    void *out;            // (B) summed error, or NULL
    double *res;          // (B, n) weighted residual of every constraint, or NULL
    int64_t B, ld;
    int32_t n, nch, L;
};

// The weighted residual of constraint c for one candidate; `channel(row)` yields the candidate's pose channel of
// that row of the fused keyframe matrices (rows of constraint c start at woff[c]).  Shared by the VALU kernel (a dot
// product per channel) and the MFMA kernel (channels already in LDS), so both produce the same value.
// Product of the chain's m quaternions (rows r_q ..), each normalised like transformations.quaternion_matrix does.
template <typename ChannelFn>
__device__ __forceinline__ void mg_chain_orientation(ChannelFn channel, int r_q, int m, double (&q)[4]) {
    double aw = 1.0, ax = 0.0, ay = 0.0, az = 0.0;
    for (int i = 0; i < m; i++) {
        double qw = channel(r_q + 4 * i), qx = channel(r_q + 1 + 4 * i), qy = channel(r_q + 2 + 4 * i), qz = channel(r_q + 3 + 4 * i);
        const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
        qw *= inv; qx *= inv; qy *= inv; qz *= inv;
        const double nw = aw * qw - ax * qx - ay * qy - az * qz, nx = aw * qx + ax * qw + ay * qz - az * qy;
        const double ny = aw * qy - ax * qz + ay * qw + az * qx, nz = aw * qz + ax * qy - ay * qx + az * qw;
        aw = nw; ax = nx; ay = ny; az = nz;
    }
    q[0] = aw; q[1] = ax; q[2] = ay; q[3] = az;
}
// v' = v + 2 w (u x v) + 2 u x (u x v) for a unit quaternion (w, u)
__device__ __forceinline__ void mg_rotate(const double (&q)[4], double vx, double vy, double vz, double (&out)[3]) {
    const double cx = q[2] * vz - q[3] * vy, cy = q[3] * vx - q[1] * vz, cz = q[1] * vy - q[2] * vx;
    const double dx = q[2] * cz - q[3] * cy, dy = q[3] * cx - q[1] * cz, dz = q[1] * cy - q[2] * cx;
    out[0] = vx + 2.0 * (q[0] * cx + dx);
    out[1] = vy + 2.0 * (q[0] * cy + dy);
    out[2] = vz + 2.0 * (q[0] * cz + dz);
}
// Forward kinematics: p = root translation (rows r_p ..) + sum_i R(q_0 .. q_i) offset_i over the chain's m links
// (quaternion rows r_q .., offsets off[m][3]).
template <typename ChannelFn>
__device__ __forceinline__ void mg_fk_position(ChannelFn channel, int r_p, int r_q, int m, const double *off, double (&p)[3]) {
    double p0 = channel(r_p), p1 = channel(r_p + 1), p2 = channel(r_p + 2);
    double aw = 1.0, ax = 0.0, ay = 0.0, az = 0.0;   // accumulated global rotation of the parent
    for (int i = 0; i < m; i++) {
        double qw = channel(r_q + 4 * i), qx = channel(r_q + 1 + 4 * i), qy = channel(r_q + 2 + 4 * i), qz = channel(r_q + 3 + 4 * i);
        const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
        qw *= inv; qx *= inv; qy *= inv; qz *= inv;
        const double nw = aw * qw - ax * qx - ay * qy - az * qz, nx = aw * qx + ax * qw + ay * qz - az * qy;
        const double ny = aw * qy - ax * qz + ay * qw + az * qx, nz = aw * qz + ax * qy - ay * qx + az * qw;
        aw = nw; ax = nx; ay = ny; az = nz;
        const double ox = off[3 * i], oy = off[3 * i + 1], oz = off[3 * i + 2];
        // v' = v + 2 w (u x v) + 2 u x (u x v), u = (ax, ay, az)
        const double cx = ay * oz - az * oy, cy = az * ox - ax * oz, cz = ax * oy - ay * ox;
        const double dx = ay * cz - az * cy, dy = az * cx - ax * cz, dz = ax * cy - ay * cx;
        p0 += ox + 2.0 * (aw * cx + dx);
        p1 += oy + 2.0 * (aw * cy + dy);
        p2 += oz + 2.0 * (aw * cz + dz);
    }
    p[0] = p0; p[1] = p1; p[2] = p2;
}

// Forward kinematics through a pose table record: link k rotates by the quaternion at row r0 + rec[5 + 4k] (identity if
// negative) and moves by the offset rec[6 + 4k ..]; rows of the pose block start at r0 (root xyz first).
template <typename ChannelFn>
__device__ __forceinline__ void mg_fk_position_table(ChannelFn channel, int r0, const double *rec, double (&p)[3]) {
    double p0 = channel(r0), p1 = channel(r0 + 1), p2 = channel(r0 + 2);
    double aw = 1.0, ax = 0.0, ay = 0.0, az = 0.0;
    const int m = (int)rec[4];
    for (int k = 0; k < m; k++) {
        const int qr = (int)rec[5 + 4 * k];
        if (qr >= 0) {
            double qw = channel(r0 + qr), qx = channel(r0 + qr + 1), qy = channel(r0 + qr + 2), qz = channel(r0 + qr + 3);
            const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
            qw *= inv; qx *= inv; qy *= inv; qz *= inv;
            const double nw = aw * qw - ax * qx - ay * qy - az * qz, nx = aw * qx + ax * qw + ay * qz - az * qy;
            const double ny = aw * qy - ax * qz + ay * qw + az * qx, nz = aw * qz + ax * qy - ay * qx + az * qw;
            aw = nw; ax = nx; ay = ny; az = nz;
        }
        const double ox = rec[6 + 4 * k], oy = rec[7 + 4 * k], oz = rec[8 + 4 * k];
        const double cx = ay * oz - az * oy, cy = az * ox - ax * oz, cz = ax * oy - ay * ox;
        const double dx = ay * cz - az * cy, dy = az * cx - ax * cz, dz = ax * cy - ay * cx;
        p0 += ox + 2.0 * (aw * cx + dx);
        p1 += oy + 2.0 * (aw * cy + dy);
        p2 += oz + 2.0 * (aw * cz + dz);
    }
    p[0] = p0; p[1] = p1; p[2] = p2;
}

// The candidate's 2-D aligning transform (mg_alignment_desc): rotation about y by the angle between its own heading
// in the first control point and the previous motion's, as (cos, sin) = (h . b, h x b), and the xz translation that
// puts its first root position on the previous one.
struct mg_align2d { double c, s, tx, tz, ty; };   // ty: the start-pose mode raises every position by the start height
template <typename ChannelFn>
__device__ __forceinline__ mg_align2d mg_candidate_alignment(const mg_score_args &a, ChannelFn channel, int64_t cand) {
    const double *al = a.align;
    const int r0 = a.woff[a.n], m = (int)al[0];
    if (m == 0) {
        // start-pose mode (reference objective_functions.py:38-47): the SAME rotation about y for every candidate,
        // (cos, sin) = al[1..2], the candidate's first root position moved to (al[3], ., al[4]), heights raised by al[5]
        mg_align2d t;
        t.c = al[1]; t.s = al[2];
        const double p0x = channel(r0), p0z = channel(r0 + 2);
        t.tx = al[3] - (t.c * p0x + t.s * p0z);
        t.tz = al[4] - (t.c * p0z - t.s * p0x);
        t.ty = al[5];
        return t;
    }
    double aw = 1.0, ax = 0.0, ay = 0.0, az = 0.0;   // global orientation of the aligning node
    for (int i = 0; i < m; i++) {
        double qw = channel(r0 + 3 + 4 * i), qx = channel(r0 + 4 + 4 * i), qy = channel(r0 + 5 + 4 * i), qz = channel(r0 + 6 + 4 * i);
        const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
        qw *= inv; qx *= inv; qy *= inv; qz *= inv;
        const double nw = aw * qw - ax * qx - ay * qy - az * qz, nx = aw * qx + ax * qw + ay * qz - az * qy;
        const double ny = aw * qy - ax * qz + ay * qw + az * qx, nz = aw * qz + ax * qy - ay * qx + az * qw;
        aw = nw; ax = nx; ay = ny; az = nz;
    }
    const double rx = al[5], ry = al[6], rz = al[7];
    const double cx = ay * rz - az * ry, cy = az * rx - ax * rz, cz = ax * ry - ay * rx;
    const double dx = ay * cz - az * cy, dz = ax * cy - ay * cx;
    double bx = rx + 2.0 * (aw * cx + dx), bz = rz + 2.0 * (aw * cz + dz);
    const double bn = 1.0 / sqrt(bx * bx + bz * bz);
    bx *= bn; bz *= bn;
    const double *pc = a.align_cand ? a.align_cand + cand * 4 - 1 : al;   // [1..4]: previous heading (x, z), previous root (x, z)
    mg_align2d t;
    t.c = pc[1] * bx + pc[2] * bz;
    t.s = pc[1] * bz - pc[2] * bx;
    const double p0x = channel(r0), p0z = channel(r0 + 2);
    t.tx = pc[3] - (t.c * p0x + t.s * p0z);
    t.tz = pc[4] - (t.c * p0z - t.s * p0x);
    t.ty = 0.0;
    return t;
}

// ROOT_ONLY: the set holds root position / 2-D direction constraints only (the caller has checked): the other kinds' code -- and
// the registers their forward-kinematics chains want -- stay out of a kernel that has none to spare (the optimiser's objective
// inside the LDS-resident mixture kernel, four waves per SIMD).  The two kinds compiled in are the same statements: the same bits.
template <bool ROOT_ONLY = false, typename ChannelFn>
__device__ __forceinline__ double mg_constraint_residual(const mg_score_args &a, int c, ChannelFn channel, int64_t cand = 0) {
    const double *par = a.par + (size_t)c * 8;
    const int type = (int)par[0];
    const int r0 = a.woff[c];
    mg_align2d al = {1.0, 0.0, 0.0, 0.0, 0.0};
    if (a.align) al = mg_candidate_alignment(a, channel, cand);
    if constexpr (!ROOT_ONLY) {
    if (type == MG_CONSTRAINT_VALUE_POSITION) {   // not an error: the (aligned) root position's component par[2] at the keyframe
        double pj[3] = {channel(r0), channel(r0 + 1), channel(r0 + 2)};
        if (a.align) {
            const double x = pj[0], z = pj[2];
            pj[0] = al.c * x + al.s * z + al.tx;
            pj[2] = al.c * z - al.s * x + al.tz;
            pj[1] += al.ty;
        }
        return par[1] * pj[(int)par[2]];
    }
    if (type == MG_CONSTRAINT_VALUE_HEADING) {   // the (aligned) unit heading's x (par[2] = 0) or z component: xz of the joint's global orientation applied to ref_dir
        double q[4], v[3];
        mg_chain_orientation(channel, r0, a.chain[c], q);
        mg_rotate(q, par[5], par[6], par[7], v);
        double hx = v[0], hz = v[2];
        if (a.align) { hx = al.c * v[0] + al.s * v[2]; hz = al.c * v[2] - al.s * v[0]; }
        const double inv = 1.0 / sqrt(hx * hx + hz * hz);
        return par[1] * ((int)par[2] == 0 ? hx : hz) * inv;
    }
    if (type == MG_CONSTRAINT_JOINT_POSITION || type == MG_CONSTRAINT_JOINT_MIDPOINT) {
        // forward kinematics along the chain: p = t_root + sum_i R(q_0 .. q_(i-1)) offset_i, unit quaternions (w,x,y,z)
        const int m = a.chain[c] & 0xffff;
        double pj[3];
        mg_fk_position(channel, r0, r0 + 3, m, a.choff + (size_t)c * 2 * MG_MAX_CHAIN * 3, pj);
        if (type == MG_CONSTRAINT_JOINT_MIDPOINT) {   // two_hand_constraint.py:71: centre of the two joints
            double pk[3];
            mg_fk_position(channel, r0, r0 + 3 + 4 * (m > 1 ? m : 1), a.chain[c] >> 16, a.choff + ((size_t)c * 2 + 1) * MG_MAX_CHAIN * 3, pk);
#pragma unroll
            for (int i = 0; i < 3; i++) pj[i] = pj[i] + 0.5 * (pk[i] - pj[i]);
        }
        if (a.align) {
            const double x = pj[0], z = pj[2];
            pj[0] = al.c * x + al.s * z + al.tx;
            pj[2] = al.c * z - al.s * x + al.tz;
            pj[1] += al.ty;
        }
        double ds = 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double t = par[2 + i];
            if (t == t) ds += (t - pj[i]) * (t - pj[i]);
        }
        return par[1] * sqrt(ds);
    }
    if (type == MG_CONSTRAINT_POSE) {
        // pose_constraint.py:48-67: cloud of joint positions, optimal weighted 2-D fit onto the wanted cloud (Kovar et
        // al.), mean distance after the fit, + the velocity term of the first joint
        const double *tb = a.pose + (size_t)par[2];
        const int N = (int)tb[0], block = (int)tb[5];
        double sw = 0.0, sax = 0.0, saz = 0.0, sbx = 0.0, sbz = 0.0, num = 0.0, den = 0.0;
        for (int i = 0; i < N; i++) {
            const double *rec = tb + MG_POSE_HDR + (size_t)i * MG_POSE_REC;
            double b[3];
            mg_fk_position_table(channel, r0, rec, b);
            if (a.align) { const double x = b[0], z = b[2]; b[0] = al.c * x + al.s * z + al.tx; b[2] = al.c * z - al.s * x + al.tz; b[1] += al.ty; }
            const double w = rec[3];
            num += w * (rec[0] * b[2] - b[0] * rec[2]);
            den += w * (rec[0] * b[0] + rec[2] * b[2]);
            sw += w; sax += w * rec[0]; saz += w * rec[2]; sbx += w * b[0]; sbz += w * b[2];
        }
        num -= (sax * sbz - sbx * saz) / sw;
        den -= (sax * sbx + saz * sbz) / sw;
        const double theta = atan2(num, den), ct = cos(theta), st = sin(theta);
        const double x0 = (sax - sbx * ct - sbz * st) / sw, z0 = (saz + sbx * st - sbz * ct) / sw;
        double dist = 0.0, first[3] = {0.0, 0.0, 0.0};
        for (int i = 0; i < N; i++) {
            const double *rec = tb + MG_POSE_HDR + (size_t)i * MG_POSE_REC;
            double b[3];
            mg_fk_position_table(channel, r0, rec, b);
            if (a.align) { const double x = b[0], z = b[2]; b[0] = al.c * x + al.s * z + al.tx; b[2] = al.c * z - al.s * x + al.tz; b[1] += al.ty; }
            if (i == 0) { first[0] = b[0]; first[1] = b[1]; first[2] = b[2]; }
            const double bx = b[0] * ct + b[2] * st + x0, bz = b[2] * ct - b[0] * st + z0;
            const double ex = rec[0] - bx, ey = rec[1] - b[1], ez = rec[2] - bz;
            dist += sqrt(ex * ex + ey * ey + ez * ez);
        }
        double err = dist / (double)N;
        if (tb[1] != 0.0) {
            double nx[3];
            mg_fk_position_table(channel, r0 + block, tb + MG_POSE_HDR, nx);   // the first joint one frame later
            if (a.align) { const double x = nx[0], z = nx[2]; nx[0] = al.c * x + al.s * z + al.tx; nx[2] = al.c * z - al.s * x + al.tz; nx[1] += al.ty; }
            const double vx = tb[2] - (nx[0] - first[0]), vy = tb[3] - (nx[1] - first[1]), vz = tb[4] - (nx[2] - first[2]);
            err += sqrt(vx * vx + vy * vy + vz * vz);
        }
        return par[1] * err;
    }
    if (type == MG_CONSTRAINT_LOOK_AT) {
        // look_at_constraint.py:55-66: angle between where the joint looks and where the target is
        const int m = a.chain[c];
        double pj[3], q[4], v[3];
        mg_fk_position(channel, r0, r0 + 3, m - 1, a.choff + (size_t)c * 2 * MG_MAX_CHAIN * 3, pj);
        mg_chain_orientation(channel, r0 + 3, m, q);
        mg_rotate(q, par[5], par[6], par[7], v);
        if (a.align) {
            const double x = pj[0], z = pj[2], vx = v[0], vz = v[2];
            pj[0] = al.c * x + al.s * z + al.tx;
            pj[2] = al.c * z - al.s * x + al.tz;
            pj[1] += al.ty;
            v[0] = al.c * vx + al.s * vz;
            v[2] = al.c * vz - al.s * vx;
        }
        const double tx = par[2] - pj[0], ty = par[3] - pj[1], tz = par[4] - pj[2];
        const double dot = (v[0] * tx + v[1] * ty + v[2] * tz) / (sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) * sqrt(tx * tx + ty * ty + tz * tz));
        return par[1] * acos(fmin(1.0, fmax(dot, -1.0)));
    }
    if (type == MG_CONSTRAINT_JOINT_ORIENTATION) {
        // global_transform_constraint.py:109-121: angle between the joint's global orientation applied to ref_dir and the target vector
        double q[4];
        mg_chain_orientation(channel, r0, a.chain[c], q);
        double v[3];
        mg_rotate(q, par[5], par[6], par[7], v);
        if (a.align) {
            const double x = v[0], z = v[2];
            v[0] = al.c * x + al.s * z;
            v[2] = al.c * z - al.s * x;
        }
        const double dot = (v[0] * par[2] + v[1] * par[3] + v[2] * par[4]) /
                           (sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) * sqrt(par[2] * par[2] + par[3] * par[3] + par[4] * par[4]));
        return par[1] * acos(fmin(1.0, fmax(dot, -1.0)));
    }
    }   // !ROOT_ONLY
    if (type == MG_CONSTRAINT_POSITION) {
        // _point_distance: axes whose target is NaN (the reference's None) are ignored
        double ds = 0.0;
        if (a.align) {
            const double x = channel(r0), z = channel(r0 + 2);
            const double pj[3] = {al.c * x + al.s * z + al.tx, channel(r0 + 1) + al.ty, al.c * z - al.s * x + al.tz};
#pragma unroll
            for (int i = 0; i < 3; i++) {
                double t = par[2 + i];
                if (t == t) ds += (t - pj[i]) * (t - pj[i]);
            }
            return par[1] * sqrt(ds);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double t = par[2 + i];
            if (t == t) {
                const double v = channel(r0 + i);
                ds += (t - v) * (t - v);
            }
        }
        return par[1] * sqrt(ds);
    }
    // heading = xz of (rotation of the root quaternion (w,x,y,z)) applied to ref_dir
    const double qw = channel(r0 + 3), qx = channel(r0 + 4), qy = channel(r0 + 5), qz = channel(r0 + 6);
    const double nq = qw * qw + qx * qx + qy * qy + qz * qz, s2 = 2.0 / nq;
    const double rx = par[5], ry = par[6], rz = par[7];
    const double lx = (1.0 - s2 * (qy * qy + qz * qz)) * rx + s2 * (qx * qy - qz * qw) * ry + s2 * (qx * qz + qy * qw) * rz;
    const double lz = s2 * (qx * qz - qy * qw) * rx + s2 * (qy * qz + qx * qw) * ry + (1.0 - s2 * (qx * qx + qy * qy)) * rz;
    const double px = a.align ? al.c * lx + al.s * lz : lx, pz = a.align ? al.c * lz - al.s * lx : lz;
    // par[2..4]: the unit target (tx, tz) = target / |target| and sqrt(tx^2 + tz^2), made on the host by these very statements
    // (mg_constraint_par_row): tn = sqrt(t0^2 + t1^2), tx = t0 / tn, tz = t1 / tn
    const double tx = par[2], tz = par[3];
    const double mn = sqrt(px * px + pz * pz);
    const double mx = px / mn, mz = pz / mn;
    double cosang = (tx * mx + tz * mz) / (par[4] * sqrt(mx * mx + mz * mz));
    cosang = fmin(1.0, fmax(cosang, -1.0));
    return par[1] * fabs(acos(cosang) * (180.0 / M_PI));
}

// (value, index) pairs of the first-minimum rule: the smaller value wins, ties go to the smaller index -- a total order, so
// partial results can be combined in any order
__device__ __forceinline__ void mg_min_combine(double &v, int64_t &i, double ov, int64_t oi) {
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}


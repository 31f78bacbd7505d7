// Fused candidate scoring for gfx950 (MI355X), float64 arithmetic, no frames materialised:
//   error_b = sum_c weight_c * constraint_c( root pose of candidate b at keyframe t_c )
// replacing the per-sample loop of evaluate_samples_using_constraints
// (reference morphablegraphs/motion_generator/motion_primitive_generator.py:230-261) over
// MotionPrimitiveConstraints.evaluate (reference .../constraints/motion_primitive_constraints.py:100-122)
// for root-joint position / 2-D direction constraints, and its first-minimum argmin.
#include "mg_internal.h"
#include <cstring>
#include "mg_gmm_device.h"

struct mg_score_args {
    const double *W;      // [rows][L]     sum_j w_j E'[(i0+j) D + d]; constraint c owns rows woff[c] ..
    const double *bias;   // [rows]        mean frame at t_c
    const double *par;    // [n][8]        type, weight, target[3], ref_dir[3]
    const int32_t *woff;  // [n + 1]
    const int32_t *chain; // [n]           FK chain length
    const double *choff;  // [n][2][MG_MAX_CHAIN][3]
    const double *pose;   // pose constraints' tables (MG_POSE_HDR / MG_POSE_REC layout) or NULL
    const double *align;  // [8] or NULL: chain length, previous heading (x,z), previous root (x,z), ref_dir; rows at woff[n]
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
__device__ __forceinline__ mg_align2d mg_candidate_alignment(const mg_score_args &a, ChannelFn channel) {
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
    mg_align2d t;
    t.c = al[1] * bx + al[2] * bz;
    t.s = al[1] * bz - al[2] * bx;
    const double p0x = channel(r0), p0z = channel(r0 + 2);
    t.tx = al[3] - (t.c * p0x + t.s * p0z);
    t.tz = al[4] - (t.c * p0z - t.s * p0x);
    t.ty = 0.0;
    return t;
}

template <typename ChannelFn>
__device__ __forceinline__ double mg_constraint_residual(const mg_score_args &a, int c, ChannelFn channel) {
    const double *par = a.par + (size_t)c * 8;
    const int type = (int)par[0];
    const int r0 = a.woff[c];
    mg_align2d al = {1.0, 0.0, 0.0, 0.0, 0.0};
    if (a.align) al = mg_candidate_alignment(a, channel);
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
    const double tn = sqrt(par[2] * par[2] + par[3] * par[3]);
    const double tx = par[2] / tn, tz = par[3] / tn;
    const double mn = sqrt(px * px + pz * pz);
    const double mx = px / mn, mz = pz / mn;
    double cosang = (tx * mx + tz * mz) / (sqrt(tx * tx + tz * tz) * sqrt(mx * mx + mz * mz));
    cosang = fmin(1.0, fmax(cosang, -1.0));
    return par[1] * fabs(acos(cosang) * (180.0 / M_PI));
}

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
        lds_r[c * MG_SC_CANDS + lane] = mg_constraint_residual(a, c, channel);
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
        resid[c * 16 + cand] = mg_constraint_residual(a, c, [&](int row) { return v[row]; });
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

template <int KK>
static int mg_launch_score_mfma_kk(mg_primitive *p, const mg_constraint_set *cs, const mg_score_args &a, bool lf, bool of) {
    const size_t lds = (size_t)4 * (16 * (cs->RT * 16 + 1) + (size_t)std::max(cs->n, 1) * 16) * 8;
    if (lds > 150 * 1024) return MG_ERR_UNSUPPORTED;
    const int64_t grid = (a.B + 63) / 64;
    if (grid > 0x7fffffff) return MG_ERR_UNSUPPORTED;
    hipStream_t st = p->ctx->stream;
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

int mg_launch_score(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int ldt, int64_t B, int64_t ld, void *out, int odt, double *res) {
    mg_score_args a;
    a.res = res;
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
__device__ __forceinline__ void mg_min_combine(double &v, int64_t &i, double ov, int64_t oi) {
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}

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
__global__ __launch_bounds__(1024) void mg_argmin_gather_kernel(const void *vals, int64_t n, void *out, const void *x, int x_f64, int64_t ld, int L) {
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
        ((int64_t *)out)[0] = bi;
        ((double *)out)[1] = best;
        winner = bi;
    }
    __syncthreads();
    const int64_t idx = winner;
    double *row = (double *)((char *)out + 16);
    for (int i = tid; i < L; i += 1024)
        row[i] = x_f64 ? ((const double *)x)[idx * ld + i] : (double)((const float *)x)[idx * ld + i];
}
int mg_launch_argmin_gather(mg_context *ctx, const void *v, int dt, int64_t n, void *result_dev, const void *x, int xdt, int64_t ld, int L) {
    if (dt == MG_F64) hipLaunchKernelGGL((mg_argmin_gather_kernel<true>), dim3(1), dim3(1024), 0, ctx->stream, v, n, result_dev, x, xdt == MG_F64 ? 1 : 0, ld, L);
    else hipLaunchKernelGGL((mg_argmin_gather_kernel<false>), dim3(1), dim3(1024), 0, ctx->stream, v, n, result_dev, x, xdt == MG_F64 ? 1 : 0, ld, L);
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

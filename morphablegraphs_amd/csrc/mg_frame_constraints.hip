// Constraints that walk a joint through EVERY frame of a candidate's motion, on the device (gfx950).  Reference
// morphablegraphs/constraints/spatial_constraints/: keyframe_constraints/global_transform_ca_constraint.py:33-46 (the closest a
// joint ever comes to a point), discrete_trajectory_constraint.py:66-90 (frame by frame against a list of points),
// keyframe_constraints/local_trajectory_constraint.py:45-78 and trajectory_set_constraint.py:82-104 (targets looked up on a
// Catmull-Rom spline BY THE ARC LENGTH the joint has walked), keyframe_constraints/joint_rotation_constraint.py:55-72 (a joint's
// local rotation in one frame).  The TrajectoryConstraint itself (closest point per frame) is mg_score_trajectory_points.
//
// Inputs are what the library already makes on the device: float64 frames of the whole batch (mg_back_project_frames_f64), turned
// and moved onto the previous motion per candidate by mg_align_frames, and the joints' tracks (mg_joint_positions).  One lane
// per candidate: every constraint here is a chain over the frames (a running arc length, a running minimum), a few hundred
// flops per frame; (B, T, J, 3) float64 tracks in, 8 bytes (and optionally the residual vector) per candidate out.
// Oracle: oracle/mg_oracle.py per_frame_constraint_residuals.  PARITY UNPINNED where anim_utils' forward kinematics is involved;
// the arc-length look-up is pinned (tests/golden/trajectory_spline.npz through the oracle's restatement).
#include <cmath>

#include "mg_internal.h"

#define MG_FC_BLOCK 64

// ---------------------------------------------------------------------------------------------------------------------
// mg_align_frames: every candidate's frames turned about y and moved in xz (and lifted, for a start pose) like the fused
// scorer aligns a candidate (mg_candidate_alignment): the transform comes from the candidate's own first control point --
// vals (B, n_vals) = its root position (x, z) and, when aligning to a previous frame, its heading (x, z), the residuals of
// MG_CONSTRAINT_VALUE_POSITION / _HEADING constraints at t = 0 (mg_score_constraint_residuals).  Rotating the control points
// and evaluating the spline (the reference, motion_primitive_constraints.py:106-116) or evaluating and rotating the frame is the
// same map: the spline is linear in its control points.
// ---------------------------------------------------------------------------------------------------------------------
struct mg_align_args {
    double *frames;
    const double *vals;
    int64_t B;
    int32_t T, D, n_vals, start_pose;
    double h0, h1, px, py, pz;
};

__global__ __launch_bounds__(256) void mg_align_frames_kernel(mg_align_args a) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.B * a.T) return;
    const int64_t b = e / a.T;
    const double *v = a.vals + b * a.n_vals;
    const double p0x = v[0], p0z = v[1];
    double c, s, ty;
    if (a.start_pose) {
        c = a.h0; s = a.h1; ty = a.py;
    } else {
        const double bx = v[2], bz = v[3];
        c = a.h0 * bx + a.h1 * bz;
        s = a.h0 * bz - a.h1 * bx;
        ty = 0.0;
    }
    const double tx = a.px - (c * p0x + s * p0z), tz = a.pz - (c * p0z - s * p0x);
    double *f = a.frames + e * a.D;
    const double x = f[0], z = f[2];
    f[0] = c * x + s * z + tx;
    f[1] += ty;
    f[2] = c * z - s * x + tz;
    if (a.D >= 7) {   // the root's quaternion turns with the candidate: (cos(phi / 2), 0, sin(phi / 2), 0) x q
        const double phi = atan2(s, c);
        const double aw = cos(0.5 * phi), ay = sin(0.5 * phi);
        const double qw = f[3], qx = f[4], qy = f[5], qz = f[6];
        f[3] = aw * qw - ay * qy;
        f[4] = aw * qx + ay * qz;
        f[5] = aw * qy + ay * qw;
        f[6] = aw * qz - ay * qx;
    }
}

extern "C" int mg_align_frames(mg_primitive *p, double *frames_dev, int64_t B, int32_t T, const double *vals_dev, int32_t n_vals,
                               const mg_alignment_desc *al) {
    if (!p || !al || B < 0 || T < 1 || (n_vals != 2 && n_vals != 4)) {
        mg_set_error("mg_align_frames: bad arguments (n_vals is 2 for a start pose, 4 for a previous frame)");
        return MG_ERR_INVALID_ARGUMENT;
    }
    const bool start_pose = al->joint == MG_ALIGN_START_POSE;
    if (start_pose ? n_vals < 2 : n_vals < 4) { mg_set_error("mg_align_frames: aligning to a previous frame needs the candidates' headings (n_vals 4)"); return MG_ERR_INVALID_ARGUMENT; }
    const double hn = std::sqrt(al->heading[0] * al->heading[0] + al->heading[1] * al->heading[1]);
    if (!(hn > 0.0) || !std::isfinite(hn)) { mg_set_error("mg_align_frames: heading is zero or not finite"); return MG_ERR_INVALID_ARGUMENT; }
    if (B == 0) return MG_OK;
    if (!frames_dev || !vals_dev) { mg_set_error("mg_align_frames: NULL pointer"); return MG_ERR_INVALID_ARGUMENT; }
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    mg_align_args a;
    a.frames = frames_dev; a.vals = vals_dev; a.B = B; a.T = T; a.D = (int32_t)p->D; a.n_vals = n_vals; a.start_pose = start_pose ? 1 : 0;
    a.h0 = al->heading[0] / hn; a.h1 = al->heading[1] / hn; a.px = al->position[0]; a.py = al->position[1]; a.pz = al->position[2];
    const int64_t n = B * T;
    hipLaunchKernelGGL(mg_align_frames_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->ctx->stream, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// mg_score_frame_constraint
// ---------------------------------------------------------------------------------------------------------------------
struct mg_fc_traj { const double *poly, *arc; double full; int32_t n_seg, G; };
struct mg_fc_args {
    const double *tracks;        // (B, T, J, 3), or (B, T, D) frames for MG_FRAME_JOINT_ROTATION
    double *out, *res;
    const double *points;
    int64_t B;
    int32_t T, J, type, nf, n_points, accumulate, quat_channel;
    double weight, start_arc;
    double target[3];
    int32_t axis_on[3];
    double quat[4];
    mg_fc_traj traj[MG_FRAME_MAX_JOINTS];
    double arc0[MG_FRAME_MAX_JOINTS], range_start[MG_FRAME_MAX_JOINTS], range_end[MG_FRAME_MAX_JOINTS];
    int32_t has_range[MG_FRAME_MAX_JOINTS];
};

__device__ __forceinline__ void mg_fc_point(const mg_fc_traj &t, double u, double *p) {
    const double scaled = t.n_seg * u;
    int index = (int)floor(scaled);
    if (index >= t.n_seg) {
        const double *q = t.poly + (size_t)t.n_seg * 12;
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
        return;
    }
    const double tt = scaled - index;
    const double *A = t.poly + (size_t)index * 12;
#pragma unroll
    for (int d = 0; d < 3; d++) p[d] = ((A[d] * tt + A[3 + d]) * tt + A[6 + d]) * tt + A[9 + d];
}

// ParameterizedSpline.query_point_by_absolute_arc_length (splines/parameterized_spline.py:131-155): beyond the full arc length the last
// control point; else the table entries bounding the relative arc length, their parameters interpolated linearly
// (arc_length_map.py:97-160)
__device__ __forceinline__ void mg_fc_point_by_arc(const mg_fc_traj &t, double arc, double *p) {
    if (arc > t.full || !(t.full > 0.0)) {
        const double *q = t.poly + (size_t)t.n_seg * 12;
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
        return;
    }
    const double rel = arc / t.full;
    double u;
    if (rel <= t.arc[0]) {
        u = 0.0;
    } else if (rel >= t.arc[t.G]) {
        u = 1.0;
    } else {
        int lo = 0, hi = t.G;                       // arc[lo] <= rel < arc[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (t.arc[mid] <= rel) lo = mid; else hi = mid;
        }
        const double l0 = t.arc[lo], l1 = t.arc[lo + 1];
        const double u0 = lo / (double)t.G, u1 = (lo + 1) / (double)t.G;
        u = l0 == rel ? u0 : u0 + (rel - l0) / (l1 - l0) * (u1 - u0);
    }
    mg_fc_point(t, u, p);
}

__device__ __forceinline__ void mg_fc_rotmat(const double *q, double *m) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = 1.0 - 2.0 * (y * y + z * z); m[1] = 2.0 * (x * y - z * w); m[2] = 2.0 * (x * z + y * w);
    m[3] = 2.0 * (x * y + z * w); m[4] = 1.0 - 2.0 * (x * x + z * z); m[5] = 2.0 * (y * z - x * w);
    m[6] = 2.0 * (x * z - y * w); m[7] = 2.0 * (y * z + x * w); m[8] = 1.0 - 2.0 * (x * x + y * y);
}

__global__ __launch_bounds__(MG_FC_BLOCK) void mg_frame_constraint_kernel(mg_fc_args a) {
    const int64_t b = (int64_t)blockIdx.x * MG_FC_BLOCK + threadIdx.x;
    if (b >= a.B) return;
    const int T = a.T, J = a.J;
    const double *tr = a.tracks + b * (int64_t)T * J * 3;
    double err = 0.0;
    if (a.type == MG_FRAME_CA_POSITION) {
        // errors[i] = _point_distance(position, joint position in frame i); error = min (global_transform_ca_constraint.py:33-39)
        double best = INFINITY;
        for (int f = 0; f < a.nf; f++) {
            double d2 = 0.0;
#pragma unroll
            for (int d = 0; d < 3; d++)
                if (a.axis_on[d]) { const double v = a.target[d] - tr[(int64_t)f * 3 + d]; d2 += v * v; }
            const double dist = sqrt(d2);
            best = dist < best ? dist : best;
        }
        err = a.weight * best;
        if (a.res) a.res[b] = err;
    } else if (a.type == MG_FRAME_DISCRETE_TRAJECTORY) {
        // per frame the distance to the list's point of the same index, unconstrained axes zeroed, frames beyond the list 0;
        // error = the average over ALL frames (discrete_trajectory_constraint.py:66-90)
        double sum = 0.0;
        for (int f = 0; f < T; f++) {
            double r = 0.0;
            if (f < a.n_points) {
                double d2 = 0.0;
#pragma unroll
                for (int d = 0; d < 3; d++)
                    if (a.axis_on[d]) { const double v = tr[(int64_t)f * 3 + d] - a.points[(int64_t)f * 3 + d]; d2 += v * v; }
                r = sqrt(d2);
            }
            sum += r;
            if (a.res) a.res[b * T + f] = a.weight * r;
        }
        err = a.weight * (sum / T);
    } else if (a.type == MG_FRAME_LOCAL_TRAJECTORY) {
        // the arc length walked so far picks the target on the spline; squared xz distance, summed (local_trajectory_constraint.py:61-78)
        double arc = a.start_arc, sum = 0.0, last[3] = {0.0, 0.0, 0.0};
        for (int f = 0; f < a.nf; f++) {
            const double *p = tr + (int64_t)f * 3;
            if (f > 0) arc += sqrt((last[0] - p[0]) * (last[0] - p[0]) + (last[1] - p[1]) * (last[1] - p[1]) + (last[2] - p[2]) * (last[2] - p[2]));
            double tg[3];
            mg_fc_point_by_arc(a.traj[0], arc, tg);
            const double dx = tg[0] - p[0], dz = tg[2] - p[2];
            const double r = dx * dx + dz * dz;
            sum += r;
            if (a.res) a.res[b * a.nf + f] = a.weight * r;
            last[0] = p[0]; last[1] = p[1]; last[2] = p[2];
        }
        err = a.weight * sum;
    } else if (a.type == MG_FRAME_TRAJECTORY_SET) {
        // trajectory_set_constraint.py:82-104: per frame, if any trajectory is active at its joint's arc length, the distance between
        // the mean of ALL components of the joints' positions and the mean of all components of their targets (np.average over a list
        // of points is one number); a frame's step is added to the arc lengths AFTER the frame has been looked up, from the second on
        double arcs[MG_FRAME_MAX_JOINTS], last[MG_FRAME_MAX_JOINTS][3];
        for (int j = 0; j < J; j++) arcs[j] = a.arc0[j];
        double sum = 0.0;
        for (int f = 0; f < a.nf; f++) {
            const double *p = tr + (int64_t)f * J * 3;
            bool active = false;
            for (int j = 0; j < J; j++) active = active || (a.has_range[j] && a.range_start[j] <= arcs[j] && arcs[j] <= a.range_end[j]);
            double r = 0.0;
            if (active) {
                double actual = 0.0, target = 0.0;
                for (int j = 0; j < J; j++) {
                    double tg[3];
                    mg_fc_point_by_arc(a.traj[j], arcs[j], tg);
                    actual += p[j * 3] + p[j * 3 + 1] + p[j * 3 + 2];
                    target += tg[0] + tg[1] + tg[2];
                }
                r = fabs(actual / (3.0 * J) - target / (3.0 * J));
            }
            if (f > 0)
                for (int j = 0; j < J; j++) {
                    const double dx = p[j * 3] - last[j][0], dy = p[j * 3 + 1] - last[j][1], dz = p[j * 3 + 2] - last[j][2];
                    arcs[j] += sqrt(dx * dx + dy * dy + dz * dz);
                }
            for (int j = 0; j < J; j++) { last[j][0] = p[j * 3]; last[j][1] = p[j * 3 + 1]; last[j][2] = p[j * 3 + 2]; }
            sum += r;
            if (a.res) a.res[b * a.nf + f] = a.weight * r;
        }
        err = a.weight * (sum / a.nf);
    } else {   // MG_FRAME_JOINT_ROTATION: tracks = frames (B, T, J = n_dim), the first one read
        // the Frobenius norm of the difference of the wanted and the joint's LOCAL rotation matrix (joint_rotation_constraint.py:55-72)
        const double *q = a.tracks + b * (int64_t)T * J + a.quat_channel;
        const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const double qn[4] = {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
        double m0[9], m1[9], s2 = 0.0;
        mg_fc_rotmat(a.quat, m0);
        mg_fc_rotmat(qn, m1);
#pragma unroll
        for (int e = 0; e < 9; e++) s2 += (m0[e] - m1[e]) * (m0[e] - m1[e]);
        err = a.weight * sqrt(s2);
        if (a.res) a.res[b] = err;
    }
    a.out[b] = a.accumulate ? a.out[b] + err : err;
}

extern "C" int mg_frame_constraint_width(const mg_frame_constraint_desc *c, int32_t n_times) {
    if (!c) return 0;
    switch (c->type) {
        case MG_FRAME_CA_POSITION: case MG_FRAME_JOINT_ROTATION: return 1;
        case MG_FRAME_DISCRETE_TRAJECTORY: return n_times;
        case MG_FRAME_LOCAL_TRAJECTORY: case MG_FRAME_TRAJECTORY_SET: return c->n_frames > 0 && c->n_frames < n_times ? c->n_frames : n_times;
        default: return 0;
    }
}

extern "C" int mg_score_frame_constraint(mg_primitive *p, const mg_frame_constraint_desc *c, const double *tracks_dev, int64_t B, int32_t T,
                                         int32_t J, double *errors_dev, int accumulate, double *residuals_dev) {
    if (!p || !c || B < 0 || T < 1 || J < 1 || !std::isfinite(c->weight)) { mg_set_error("mg_score_frame_constraint: bad arguments"); return MG_ERR_INVALID_ARGUMENT; }
    mg_fc_args a = {};
    a.tracks = tracks_dev; a.out = errors_dev; a.res = residuals_dev; a.B = B; a.T = T; a.J = J; a.type = c->type; a.accumulate = accumulate ? 1 : 0;
    a.weight = c->weight; a.start_arc = c->start_arc; a.points = c->points_dev; a.n_points = c->n_points; a.quat_channel = c->quat_channel;
    a.nf = c->n_frames > 0 && c->n_frames < T ? c->n_frames : T;
    for (int d = 0; d < 3; d++) { a.target[d] = c->target[d]; a.axis_on[d] = c->axis_on[d] ? 1 : 0; }
    int n_traj = 0;
    switch (c->type) {
        case MG_FRAME_CA_POSITION:
            if (J != 1) { mg_set_error("mg_score_frame_constraint: one joint's track expected"); return MG_ERR_INVALID_ARGUMENT; }
            for (int d = 0; d < 3; d++)
                if (a.axis_on[d] && !std::isfinite(a.target[d])) { mg_set_error("mg_score_frame_constraint: target axis %d is not finite", d); return MG_ERR_INVALID_ARGUMENT; }
            break;
        case MG_FRAME_DISCRETE_TRAJECTORY:
            if (J != 1 || c->n_points < 0 || (c->n_points > 0 && !c->points_dev)) { mg_set_error("mg_score_frame_constraint: discrete trajectory needs one track and its points on the device"); return MG_ERR_INVALID_ARGUMENT; }
            break;
        case MG_FRAME_LOCAL_TRAJECTORY:
            if (J != 1) { mg_set_error("mg_score_frame_constraint: one joint's track expected"); return MG_ERR_INVALID_ARGUMENT; }
            n_traj = 1;
            break;
        case MG_FRAME_TRAJECTORY_SET:
            if (J > MG_FRAME_MAX_JOINTS || c->n_joints != J) { mg_set_error("mg_score_frame_constraint: a trajectory set takes 1..%d joints, one track each", MG_FRAME_MAX_JOINTS); return MG_ERR_INVALID_ARGUMENT; }
            n_traj = J;
            break;
        case MG_FRAME_JOINT_ROTATION: {
            if (c->quat_channel < 3 || c->quat_channel + 4 > J) { mg_set_error("mg_score_frame_constraint: quaternion channel %d outside the frame (n_dim = %d)", c->quat_channel, J); return MG_ERR_INVALID_ARGUMENT; }
            const double n = std::sqrt(c->quaternion[0] * c->quaternion[0] + c->quaternion[1] * c->quaternion[1] + c->quaternion[2] * c->quaternion[2] + c->quaternion[3] * c->quaternion[3]);
            if (!(n > 0.0) || !std::isfinite(n)) { mg_set_error("mg_score_frame_constraint: rotation target is zero or not finite"); return MG_ERR_INVALID_ARGUMENT; }
            for (int e = 0; e < 4; e++) a.quat[e] = c->quaternion[e] / n;
            break;
        }
        default: mg_set_error("mg_score_frame_constraint: unknown type %d", c->type); return MG_ERR_INVALID_ARGUMENT;
    }
    for (int j = 0; j < n_traj; j++) {
        const mg_trajectory *t = c->trajectories[j];
        if (!t || t->prim != p) { mg_set_error("mg_score_frame_constraint: trajectory %d missing or of another primitive", j); return MG_ERR_INVALID_ARGUMENT; }
        a.traj[j] = {t->d_poly, t->d_arc, t->full_arc, t->n_seg, t->granularity};
        a.arc0[j] = c->arc0[j]; a.range_start[j] = c->range_start[j]; a.range_end[j] = c->range_end[j]; a.has_range[j] = c->has_range[j] ? 1 : 0;
    }
    if (B == 0) return MG_OK;
    if (!tracks_dev || !errors_dev) { mg_set_error("mg_score_frame_constraint: NULL pointer"); return MG_ERR_INVALID_ARGUMENT; }
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipLaunchKernelGGL(mg_frame_constraint_kernel, dim3((unsigned)((B + MG_FC_BLOCK - 1) / MG_FC_BLOCK)), dim3(MG_FC_BLOCK), 0, p->ctx->stream, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

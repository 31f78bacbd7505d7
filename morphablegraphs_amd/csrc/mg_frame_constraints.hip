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
#include <cstring>

#include <algorithm>
#include <new>
#include <vector>

#include "mg_internal.h"
#include "mg_score_device.h"
#include "mg_traj_device.h"

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
struct mg_fc_traj { const double *poly, *arc; double full; int32_t n_seg, G; int32_t poly_off, arc_off; };   // *_off: the table's place in LDS (doubles), TABLES_LDS kernels
// Where a trajectory's tables are read from: the copies the workgroup made in LDS (TABLES_LDS: every look-up of the walks below is a
// dependent read -- the closest-point search makes about 17 per frame, the arc-length search 10 -- so LDS latency instead of L2's is
// most of the kernel's time), or global memory when the list's tables do not fit.
template <bool TABLES_LDS> __device__ __forceinline__ const double *mg_fc_poly(const mg_fc_traj &t, const double *lds) {
    if constexpr (TABLES_LDS) return lds + t.poly_off; else return t.poly;
}
template <bool TABLES_LDS> __device__ __forceinline__ const double *mg_fc_arc(const mg_fc_traj &t, const double *lds) {
    if constexpr (TABLES_LDS) return lds + t.arc_off; else return t.arc;
}
struct mg_fc_args {
    const double *tracks;        // (B, T, J, 3), or (B, T, D) frames for MG_FRAME_JOINT_ROTATION
    double *out, *res;
    const double *points;
    int64_t B;
    int32_t T, J, type, nf, n_points, accumulate, quat_channel;
    int32_t search;              // MG_FRAME_JOINT_TRAJECTORY: 0 the reference's search, 1 the monotone walk (MG_OPT_TRAJECTORY_SEARCH)
    double weight, start_arc;
    double target[3];
    int32_t axis_on[3];
    double quat[4];
    mg_fc_traj traj[MG_FRAME_MAX_JOINTS];
    double arc0[MG_FRAME_MAX_JOINTS], range_start[MG_FRAME_MAX_JOINTS], range_end[MG_FRAME_MAX_JOINTS];
    int32_t has_range[MG_FRAME_MAX_JOINTS];
};

__device__ __forceinline__ void mg_fc_point(const mg_fc_traj &t, const double *poly, double u, double *p) {
    const double scaled = t.n_seg * u;
    int index = (int)floor(scaled);
    if (index >= t.n_seg) {
        const double *q = poly + (size_t)t.n_seg * 12;
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
        return;
    }
    const double tt = scaled - index;
    const double *A = poly + (size_t)index * 12;
#pragma unroll
    for (int d = 0; d < 3; d++) p[d] = ((A[d] * tt + A[3 + d]) * tt + A[6 + d]) * tt + A[9 + d];
}

// ParameterizedSpline.query_point_by_absolute_arc_length (splines/parameterized_spline.py:131-155): beyond the full arc length the last
// control point; else the table entries bounding the relative arc length, their parameters interpolated linearly
// (arc_length_map.py:97-160)
__device__ __forceinline__ void mg_fc_point_by_arc(const mg_fc_traj &t, const double *poly, const double *arcs, double arc, double *p) {
    if (arc > t.full || !(t.full > 0.0)) {
        const double *q = poly + (size_t)t.n_seg * 12;
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
        return;
    }
    const double rel = arc / t.full;
    double u;
    if (rel <= arcs[0]) {
        u = 0.0;
    } else if (rel >= arcs[t.G]) {
        u = 1.0;
    } else {
        int lo = 0, hi = t.G;                       // arc[lo] <= rel < arc[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (arcs[mid] <= rel) lo = mid; else hi = mid;
        }
        const double l0 = arcs[lo], l1 = arcs[lo + 1];
        const double u0 = lo / (double)t.G, u1 = (lo + 1) / (double)t.G;
        u = l0 == rel ? u0 : u0 + (rel - l0) / (l1 - l0) * (u1 - u0);
    }
    mg_fc_point(t, poly, u, p);
}

__device__ __forceinline__ void mg_fc_rotmat(const double *q, double *m) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = 1.0 - 2.0 * (y * y + z * z); m[1] = 2.0 * (x * y - z * w); m[2] = 2.0 * (x * z + y * w);
    m[3] = 2.0 * (x * y + z * w); m[4] = 1.0 - 2.0 * (x * x + z * z); m[5] = 2.0 * (y * z - x * w);
    m[6] = 2.0 * (x * z - y * w); m[7] = 2.0 * (y * z + x * w); m[8] = 1.0 - 2.0 * (x * x + y * y);
}

// one constraint for candidate b: its weighted error (and, if a.res is set, its residual vector)
template <bool TABLES_LDS>
__device__ __forceinline__ double mg_fc_evaluate(const mg_fc_args &a, const int64_t b, const double *lds) {
    const int T = a.T, J = a.J;
    const double *tr = a.tracks + b * (int64_t)T * J * 3;
    double err = 0.0;
    if (a.type == MG_FRAME_JOINT_TRAJECTORY) {
        // TrajectoryConstraint on the joint whose track this is (trajectory_constraint.py:79-121): per frame the distance to the closest
        // point of the spline at or after the previous frame's parameter; the error is the average (mg_score_trajectory_points'
        // arithmetic: mg_traj_device.h)
        const mg_fc_traj &t = a.traj[0];
        const double invG = 1.0 / (double)t.G;
        const double *poly = mg_fc_poly<TABLES_LDS>(t, lds);
        double min_u = a.start_arc, sum = 0.0;       // (start_arc carries the constraint's min_u)
        if (a.search == 0) {                         // the reference's search: this lane's frames at its own pace (mg_traj_chain)
            mg_traj_chain(poly, t.n_seg, T, min_u,
                          [&](int f, double *q) { const double *pp = tr + (int64_t)f * 3; q[0] = pp[0]; q[1] = pp[1]; q[2] = pp[2]; },
                          [&](int f, double dist, double, int) { sum += dist; if (a.res) a.res[b * T + f] = a.weight * dist; });
        } else {
            for (int f = 0; f < T; f++) {
                const double *pp = tr + (int64_t)f * 3;
                const double q[3] = {pp[0], pp[1], pp[2]};
                const double dist = mg_traj_closest_dist(poly, t.n_seg, t.G, invG, &min_u, q);
                sum += dist;
                if (a.res) a.res[b * T + f] = a.weight * dist;
            }
        }
        err = a.weight * (T > 0 ? sum / (double)T : 0.0);
    } else if (a.type == MG_FRAME_CA_POSITION) {
        // errors[i] = _point_distance(position, joint position in frame i); error = min (global_transform_ca_constraint.py:33-39)
        // (the minimum of the distances is the root of the minimum of their squares, exactly: sqrt is monotone and correctly rounded --
        // one root per candidate instead of one per frame, the same bits)
        double best2 = INFINITY;
        auto frame = [&](const double x, const double y, const double z) {
            const double p[3] = {x, y, z};
            double d2 = 0.0;
#pragma unroll
            for (int d = 0; d < 3; d++)
                if (a.axis_on[d]) { const double v = a.target[d] - p[d]; d2 += v * v; }
            best2 = d2 < best2 ? d2 : best2;
        };
        int f = 0;
        if ((((size_t)tr) & 15) == 0) {
            // two frames = six doubles = three 16-byte loads (a lane reads its own candidate's 3.7 KB: wider requests, half as many)
            const double2 *t2 = (const double2 *)tr;
#pragma unroll 2
            for (; f + 2 <= a.nf; f += 2) {
                const double2 u0 = t2[(f >> 1) * 3], u1 = t2[(f >> 1) * 3 + 1], u2 = t2[(f >> 1) * 3 + 2];
                frame(u0.x, u0.y, u1.x);
                frame(u1.y, u2.x, u2.y);
            }
        }
        for (; f < a.nf; f++) frame(tr[(int64_t)f * 3], tr[(int64_t)f * 3 + 1], tr[(int64_t)f * 3 + 2]);
        err = a.weight * sqrt(best2);
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
        const double *poly = mg_fc_poly<TABLES_LDS>(a.traj[0], lds), *arcs = mg_fc_arc<TABLES_LDS>(a.traj[0], lds);
        for (int f = 0; f < a.nf; f++) {
            const double *p = tr + (int64_t)f * 3;
            if (f > 0) arc += sqrt((last[0] - p[0]) * (last[0] - p[0]) + (last[1] - p[1]) * (last[1] - p[1]) + (last[2] - p[2]) * (last[2] - p[2]));
            double tg[3];
            mg_fc_point_by_arc(a.traj[0], poly, arcs, arc, tg);
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
                    mg_fc_point_by_arc(a.traj[j], mg_fc_poly<TABLES_LDS>(a.traj[j], lds), mg_fc_arc<TABLES_LDS>(a.traj[j], lds), arcs[j], tg);
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
    return err;
}

// the workgroup's copies of a constraint's tables (a table two trajectories share is written twice with the same values)
__device__ __forceinline__ void mg_fc_stage_tables(const mg_fc_args &a, double *lds) {
    for (int j = 0; j < MG_FRAME_MAX_JOINTS; j++) {
        const mg_fc_traj &t = a.traj[j];
        if (!t.poly) continue;
        if (t.poly_off >= 0) for (int e = threadIdx.x; e < t.n_seg * 12 + 3; e += MG_FC_BLOCK) lds[t.poly_off + e] = t.poly[e];
        if (t.arc_off >= 0) for (int e = threadIdx.x; e <= t.G; e += MG_FC_BLOCK) lds[t.arc_off + e] = t.arc[e];
    }
}

template <bool TABLES_LDS>
__global__ __launch_bounds__(MG_FC_BLOCK) void mg_frame_constraint_kernel(mg_fc_args a) {
    extern __shared__ double fc_lds[];
    if constexpr (TABLES_LDS) {
        mg_fc_stage_tables(a, fc_lds);
        __syncthreads();
    }
    const int64_t b = (int64_t)blockIdx.x * MG_FC_BLOCK + threadIdx.x;
    if (b >= a.B) return;
    const double err = mg_fc_evaluate<TABLES_LDS>(a, b, fc_lds);
    a.out[b] = a.accumulate ? a.out[b] + err : err;
}

// A LIST of per-frame constraints in one launch: candidate b's errors added up in list order -- what the same constraints give
// launched one after the other with accumulate (the first one overwriting unless `accumulate`): the same additions, the same bits.
#define MG_FC_LIST_MAX 4
struct mg_fc_list { int32_t n, accumulate; int64_t B; double *out; mg_fc_args c[MG_FC_LIST_MAX]; };
template <bool TABLES_LDS>
__global__ __launch_bounds__(MG_FC_BLOCK) void mg_frame_constraint_list_kernel(const mg_fc_list L) {
    extern __shared__ double fc_lds[];
    if constexpr (TABLES_LDS) {
        for (int i = 0; i < L.n; i++) mg_fc_stage_tables(L.c[i], fc_lds);
        __syncthreads();
    }
    const int64_t b = (int64_t)blockIdx.x * MG_FC_BLOCK + threadIdx.x;
    if (b >= L.B) return;
    double total = L.accumulate ? L.out[b] : 0.0;
    for (int i = 0; i < L.n; i++) {
        const double e = mg_fc_evaluate<TABLES_LDS>(L.c[i], b, fc_lds);
        total = (i == 0 && !L.accumulate) ? e : total + e;
    }
    L.out[b] = total;
}

#define MG_OPT_LISTS_MAX 24
struct mg_opt_wgs { int32_t n; int32_t wg0[MG_OPT_LISTS_MAX + 1]; };
// A planner step's per-frame lists, every option's in ONE launch, and the options' first minima with them (round 5; reference
// motion_generator/graph_walk_planner.py:184-226 scores every option's whole constraint list inside one step).  Workgroups
// [wg0[k], wg0[k + 1]) take option k: a lane adds its candidate's list (an empty list: nothing to add) to the error the launches before
// left in errors[k] -- the single launches' additions in their order, the same bits -- and the option's first minimum is taken right
// here: lanes -> wave -> one {value, index} per workgroup in global memory; the workgroup whose arrival is the option's last (a fence,
// a counter) combines them under the total order (smaller value, then smaller index; NaN and +inf never win; nothing finite: index 0,
// +inf -- mg_argmin_kernel's rule), writes {index, error} and copies the winning candidate behind them.  Release / acquire fences here:
// 64 workgroups per option, not the one-launch step's thousand.
struct mg_opt_list_entry {
    mg_fc_list L;
    const void *x;          // the option's candidates (B, ld)
    int64_t ld;
    int32_t x_f64, Lw;      // Lw: columns of the winner that go into the record
    char *result;           // {int64 index, float64 error, float64 latent[Lw]}
};
struct mg_min_partial { double v; int64_t i; };
template <bool TABLES_LDS>
__global__ __launch_bounds__(MG_FC_BLOCK) void mg_options_lists_kernel(const mg_opt_list_entry *__restrict__ tab, const mg_opt_wgs w, mg_min_partial *__restrict__ partials,
                                                                      int32_t *__restrict__ counters) {
    extern __shared__ double fc_lds[];
    int k = 0;
    while (k + 1 < w.n && (int)blockIdx.x >= w.wg0[k + 1]) k++;
    const mg_opt_list_entry &o = tab[k];
    const mg_fc_list &L = o.L;
    if constexpr (TABLES_LDS) {
        for (int i = 0; i < L.n; i++) mg_fc_stage_tables(L.c[i], fc_lds);
        __syncthreads();
    }
    const int lane = threadIdx.x;
    const int64_t b = ((int64_t)blockIdx.x - w.wg0[k]) * MG_FC_BLOCK + lane;
    double best = INFINITY;
    int64_t bi = INT64_MAX;
    if (b < L.B) {
        double total = L.accumulate ? L.out[b] : 0.0;
        for (int i = 0; i < L.n; i++) {
            const double e = mg_fc_evaluate<TABLES_LDS>(L.c[i], b, fc_lds);
            total = (i == 0 && !L.accumulate) ? e : total + e;
        }
        if (L.n > 0) L.out[b] = total;
        if (total < INFINITY) { best = total; bi = b; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const long long oi = __shfl_down((long long)bi, off, 64);
        mg_min_combine(best, bi, ov, (int64_t)oi);
    }
    const int nwg = w.wg0[k + 1] - w.wg0[k];
    int last = 0;
    if (lane == 0) {
        partials[blockIdx.x].v = best;
        partials[blockIdx.x].i = bi;
        __threadfence();                                              // release: the partial (and this workgroup's errors) before the count
        last = atomicAdd(&counters[k], 1) == nwg - 1 ? 1 : 0;
    }
    last = __shfl(last, 0, 64);
    if (!last) return;
    __threadfence();                                                  // acquire: every other workgroup's partial
    best = INFINITY; bi = INT64_MAX;
    for (int q = lane; q < nwg; q += 64) {
        const volatile mg_min_partial *pp = &partials[w.wg0[k] + q];
        mg_min_combine(best, bi, pp->v, pp->i);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const long long oi = __shfl_down((long long)bi, off, 64);
        mg_min_combine(best, bi, ov, (int64_t)oi);
    }
    best = __shfl(best, 0, 64);
    bi = (int64_t)__shfl((long long)bi, 0, 64);
    if (bi == INT64_MAX) { bi = 0; best = INFINITY; }
    if (lane == 0) {
        ((int64_t *)o.result)[0] = bi;
        ((double *)o.result)[1] = best;
        counters[k] = 0;                                              // ready for the next launch (stream ordered)
    }
    double *row = (double *)(o.result + 16);
    for (int i = lane; i < o.Lw; i += 64)
        row[i] = o.x_f64 ? ((const double *)o.x)[bi * o.ld + i] : (double)((const float *)o.x)[bi * o.ld + i];
}

extern "C" int mg_frame_constraint_width(const mg_frame_constraint_desc *c, int32_t n_times) {
    if (!c) return 0;
    switch (c->type) {
        case MG_FRAME_CA_POSITION: case MG_FRAME_JOINT_ROTATION: return 1;
        case MG_FRAME_DISCRETE_TRAJECTORY: case MG_FRAME_JOINT_TRAJECTORY: return n_times;
        case MG_FRAME_LOCAL_TRAJECTORY: case MG_FRAME_TRAJECTORY_SET: return c->n_frames > 0 && c->n_frames < n_times ? c->n_frames : n_times;
        default: return 0;
    }
}

// desc -> kernel arguments (validated); the output pointers are the caller's business
static int mg_fc_args_from_desc(const char *who, mg_primitive *p, const mg_frame_constraint_desc *c, const double *tracks_dev, int64_t B, int32_t T, int32_t J,
                                double *residuals_dev, mg_fc_args *out) {
    if (!p || !c || B < 0 || T < 1 || J < 1 || !std::isfinite(c->weight)) { mg_set_error("%s: bad arguments", who); return MG_ERR_INVALID_ARGUMENT; }
    mg_fc_args a = {};
    a.tracks = tracks_dev; a.out = nullptr; a.res = residuals_dev; a.B = B; a.T = T; a.J = J; a.type = c->type; a.accumulate = 0;
    a.weight = c->weight; a.start_arc = c->start_arc; a.points = c->points_dev; a.n_points = c->n_points; a.quat_channel = c->quat_channel;
    a.nf = c->n_frames > 0 && c->n_frames < T ? c->n_frames : T;
    a.search = p->ctx->opt[MG_OPT_TRAJECTORY_SEARCH] == 1 ? 1 : 0;
    for (int d = 0; d < 3; d++) { a.target[d] = c->target[d]; a.axis_on[d] = c->axis_on[d] ? 1 : 0; }
    int n_traj = 0;
    switch (c->type) {
        case MG_FRAME_JOINT_TRAJECTORY:
            if (J != 1 || !(c->start_arc >= 0.0 && c->start_arc <= 1.0)) { mg_set_error("%s: a joint trajectory takes one track and min_u (start_arc) in [0, 1]", who); return MG_ERR_INVALID_ARGUMENT; }
            n_traj = 1;
            break;
        case MG_FRAME_CA_POSITION:
            if (J != 1) { mg_set_error("%s: one joint's track expected", who); return MG_ERR_INVALID_ARGUMENT; }
            for (int d = 0; d < 3; d++)
                if (a.axis_on[d] && !std::isfinite(a.target[d])) { mg_set_error("%s: target axis %d is not finite", who, d); return MG_ERR_INVALID_ARGUMENT; }
            break;
        case MG_FRAME_DISCRETE_TRAJECTORY:
            if (J != 1 || c->n_points < 0 || (c->n_points > 0 && !c->points_dev)) { mg_set_error("%s: discrete trajectory needs one track and its points on the device", who); return MG_ERR_INVALID_ARGUMENT; }
            break;
        case MG_FRAME_LOCAL_TRAJECTORY:
            if (J != 1) { mg_set_error("%s: one joint's track expected", who); return MG_ERR_INVALID_ARGUMENT; }
            n_traj = 1;
            break;
        case MG_FRAME_TRAJECTORY_SET:
            if (J > MG_FRAME_MAX_JOINTS || c->n_joints != J) { mg_set_error("%s: a trajectory set takes 1..%d joints, one track each", who, MG_FRAME_MAX_JOINTS); return MG_ERR_INVALID_ARGUMENT; }
            n_traj = J;
            break;
        case MG_FRAME_JOINT_ROTATION: {
            if (c->quat_channel < 3 || c->quat_channel + 4 > J) { mg_set_error("%s: quaternion channel %d outside the frame (n_dim = %d)", who, c->quat_channel, J); return MG_ERR_INVALID_ARGUMENT; }
            const double n = std::sqrt(c->quaternion[0] * c->quaternion[0] + c->quaternion[1] * c->quaternion[1] + c->quaternion[2] * c->quaternion[2] + c->quaternion[3] * c->quaternion[3]);
            if (!(n > 0.0) || !std::isfinite(n)) { mg_set_error("%s: rotation target is zero or not finite", who); return MG_ERR_INVALID_ARGUMENT; }
            for (int e = 0; e < 4; e++) a.quat[e] = c->quaternion[e] / n;
            break;
        }
        default: mg_set_error("%s: unknown type %d", who, c->type); return MG_ERR_INVALID_ARGUMENT;
    }
    for (int j = 0; j < n_traj; j++) {
        const mg_trajectory *t = c->trajectories[j];
        if (!t || t->prim != p) { mg_set_error("%s: trajectory %d missing or of another primitive", who, j); return MG_ERR_INVALID_ARGUMENT; }
        a.traj[j] = {t->d_poly, t->d_arc, t->full_arc, t->n_seg, t->granularity, -1, -1};
        a.arc0[j] = c->arc0[j]; a.range_start[j] = c->range_start[j]; a.range_end[j] = c->range_end[j]; a.has_range[j] = c->has_range[j] ? 1 : 0;
    }
    *out = a;
    return MG_OK;
}

// Places in LDS for the tables the constraints' walks search (shared tables once); bytes needed, or 0 when there is nothing to stage
// or the tables do not fit (the offsets stay -1: the kernels read global memory).
#define MG_FC_LDS_MAX (150 * 1024)
static size_t mg_fc_place_tables(mg_fc_args *c, int n) {
    struct placed { const double *ptr; int32_t off; };
    placed seen[2 * MG_FC_LIST_MAX * MG_FRAME_MAX_JOINTS];
    int n_seen = 0;
    size_t doubles = 0;
    auto place = [&](const double *ptr, size_t count) {
        for (int i = 0; i < n_seen; i++) if (seen[i].ptr == ptr) return seen[i].off;
        const int32_t off = (int32_t)doubles;
        doubles += (count + 1) & ~(size_t)1;
        seen[n_seen++] = {ptr, off};
        return off;
    };
    for (int i = 0; i < n; i++) {
        const bool arcs = c[i].type == MG_FRAME_LOCAL_TRAJECTORY || c[i].type == MG_FRAME_TRAJECTORY_SET;
        for (int j = 0; j < MG_FRAME_MAX_JOINTS; j++) {
            mg_fc_traj &t = c[i].traj[j];
            if (!t.poly) continue;
            t.poly_off = place(t.poly, (size_t)t.n_seg * 12 + 3);
            t.arc_off = arcs ? place(t.arc, (size_t)t.G + 1) : -1;
        }
    }
    if (doubles == 0 || doubles * 8 > MG_FC_LDS_MAX) {
        for (int i = 0; i < n; i++) for (int j = 0; j < MG_FRAME_MAX_JOINTS; j++) c[i].traj[j].poly_off = c[i].traj[j].arc_off = -1;
        return 0;
    }
    return doubles * 8;
}
static int mg_fc_attributes(mg_context *ctx) {
    if (ctx->attr_traj & 8u) return MG_OK;
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frame_constraint_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frame_constraint_list_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_options_lists_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->attr_traj |= 8u;
    return MG_OK;
}

extern "C" int mg_score_frame_constraint(mg_primitive *p, const mg_frame_constraint_desc *c, const double *tracks_dev, int64_t B, int32_t T,
                                         int32_t J, double *errors_dev, int accumulate, double *residuals_dev) {
    mg_fc_args a;
    int rc = mg_fc_args_from_desc("mg_score_frame_constraint", p, c, tracks_dev, B, T, J, residuals_dev, &a);
    if (rc != MG_OK) return rc;
    a.out = errors_dev; a.accumulate = accumulate ? 1 : 0;
    if (B == 0) return MG_OK;
    if (!tracks_dev || !errors_dev) { mg_set_error("mg_score_frame_constraint: NULL pointer"); return MG_ERR_INVALID_ARGUMENT; }
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    const size_t lds = mg_fc_place_tables(&a, 1);
    if (lds) { rc = mg_fc_attributes(p->ctx); if (rc != MG_OK) return rc; }
    const dim3 grid((unsigned)((B + MG_FC_BLOCK - 1) / MG_FC_BLOCK));
    if (lds) hipLaunchKernelGGL(mg_frame_constraint_kernel<true>, grid, dim3(MG_FC_BLOCK), lds, p->ctx->stream, a);
    else hipLaunchKernelGGL(mg_frame_constraint_kernel<false>, grid, dim3(MG_FC_BLOCK), 0, p->ctx->stream, a);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// Several per-frame constraints of the same candidates in ONE launch (lists longer than MG_FC_LIST_MAX: one launch per group).
extern "C" int mg_score_frame_constraints(mg_primitive *p, int32_t n_constraints, const mg_frame_constraint_desc *const *constraints, const double *const *tracks_dev,
                                          const int32_t *n_times, const int32_t *n_joints, int64_t B, double *errors_dev, int accumulate,
                                          double *const *residuals_dev) {
    if (!p || n_constraints < 0 || (n_constraints > 0 && (!constraints || !tracks_dev || !n_times || !n_joints)) || B < 0) {
        mg_set_error("mg_score_frame_constraints: bad arguments");
        return MG_ERR_INVALID_ARGUMENT;
    }
    if (B > 0 && !errors_dev) { mg_set_error("mg_score_frame_constraints: errors_dev is NULL"); return MG_ERR_INVALID_ARGUMENT; }
    if (n_constraints == 0 || B == 0) return MG_OK;
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    // Groups of up to MG_FC_LIST_MAX constraints per launch, in list order.  A joint's TrajectoryConstraint below 65 536 candidates is a
    // launch of its own: mg_score_trajectory_points walks with eight lanes per candidate there (half the time of this scorer's one lane),
    // and adds the same number at the same place in the sum -- the groups around it accumulate in order.
    bool any = accumulate != 0;
    int k0 = 0;
    while (k0 < n_constraints) {
        const mg_frame_constraint_desc *c0 = constraints[k0];
        if (c0 && c0->type == MG_FRAME_JOINT_TRAJECTORY && B <= 65536 && n_joints[k0] == 1 && c0->trajectories[0] && tracks_dev[k0]) {
            const int rc = mg_score_trajectory_points(p, c0->trajectories[0], tracks_dev[k0], B, n_times[k0], c0->start_arc, c0->weight, errors_dev, any ? 1 : 0,
                                                      residuals_dev ? residuals_dev[k0] : nullptr);
            if (rc != MG_OK) return rc;
            any = true;
            k0++;
            continue;
        }
        mg_fc_list L = {};
        L.accumulate = any ? 1 : 0;
        L.B = B; L.out = errors_dev;
        while (L.n < MG_FC_LIST_MAX && k0 + L.n < n_constraints) {
            const int i = k0 + L.n;
            const mg_frame_constraint_desc *c = constraints[i];
            if (L.n > 0 && c && c->type == MG_FRAME_JOINT_TRAJECTORY && B <= 65536) break;     // (the next group's business)
            if (!tracks_dev[i]) { mg_set_error("mg_score_frame_constraints: tracks of constraint %d are NULL", i); return MG_ERR_INVALID_ARGUMENT; }
            int rc = mg_fc_args_from_desc("mg_score_frame_constraints", p, c, tracks_dev[i], B, n_times[i], n_joints[i],
                                          residuals_dev ? residuals_dev[i] : nullptr, &L.c[L.n]);
            if (rc != MG_OK) return rc;
            L.n++;
        }
        const size_t lds = mg_fc_place_tables(L.c, L.n);
        if (lds) { const int rc = mg_fc_attributes(p->ctx); if (rc != MG_OK) return rc; }
        const dim3 grid((unsigned)((B + MG_FC_BLOCK - 1) / MG_FC_BLOCK));
        mg_prof_begin(p->ctx, 9);
        if (lds) hipLaunchKernelGGL(mg_frame_constraint_list_kernel<true>, grid, dim3(MG_FC_BLOCK), lds, p->ctx->stream, L);
        else hipLaunchKernelGGL(mg_frame_constraint_list_kernel<false>, grid, dim3(MG_FC_BLOCK), 0, p->ctx->stream, L);
        mg_prof_end(p->ctx, 9);
        MG_HIP_CHECK(hipGetLastError());
        any = true;
        k0 += L.n;
    }
    return MG_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// mg_joint_tracks: the joints' global positions in every frame, for the whole batch, WITHOUT materialising frames.
//
// What the chain mg_back_project_frames_f64 -> mg_align_frames -> mg_joint_positions produces through (n, T, n_dim) float64 frames
// (808 MB for 8192 'walk' candidates; ~98 KB moved per candidate for a scorer whose algorithmic input is its 160-byte latent) comes
// from ONE launch here: a workgroup per candidate keeps the control points of the channels the wanted joints' chains READ -- root
// translation and the quaternions along the chains, a third of the pose for a hand -- in LDS (fma chains over the latents from the
// mean: the float64 frames kernel's arithmetic), derives the candidate's aligning transform from its first control point (the
// statements of MG_CONSTRAINT_VALUE_POSITION / _HEADING at t = 0 and of mg_align_frames_kernel), and for every (time, joint)
// evaluates just those channels (four taps each), turns the root, walks the chain (mg_joint_positions_kernel's loop).  The same
// operations on the same values: the tracks are the chain's, bit for bit; 24 bytes per (candidate, time, joint) leave the chip.
// A PLAN holds what does not change between calls: the joints' chain records, the channel list, per request its joints.
// ---------------------------------------------------------------------------------------------------------------------
#define MG_TRACK_MAX_REQUESTS 4
#define MG_TRACK_REC (1 + 4 * MG_MAX_CHAIN)       // chain length m, then per link (quaternion channel or -1, offset xyz): mg_joint_positions' record
struct mg_track_plan {
    mg_primitive *prim = nullptr;
    int32_t n_requests = 0, n_chan = 0, align_m = -1;        // align_m: links of the aligning node's chain (-1: plans without alignment support)
    int32_t align_joint = -1;                                // the node that chain belongs to
    int32_t req_joint0[MG_TRACK_MAX_REQUESTS + 1] = {0};     // request q owns records req_joint0[q] .. req_joint0[q + 1]
    double *d_records = nullptr;    // [n_joints_total][MG_TRACK_REC]
    double *d_align_rec = nullptr;  // [MG_TRACK_REC]: the aligning node's chain (quaternion channels only are read)
    int32_t *d_chan = nullptr;      // [n_chan] pose channels kept in LDS
    int32_t *d_slot = nullptr;      // [D] channel -> slot or -1
};

struct mg_track_args {
    const double *Et64, *mean;
    const void *lat;
    int64_t B, ld;
    int32_t L, R, D, NB, lat_f64, n_requests, n_chan, align_mode;   // align_mode 0 none, 1 previous frame, 2 start pose
    const double *records, *align_rec;
    const int32_t *chan, *slot;
    int32_t req_joint0[MG_TRACK_MAX_REQUESTS + 1], T[MG_TRACK_MAX_REQUESTS];
    const int32_t *i0[MG_TRACK_MAX_REQUESTS];
    const double *w[MG_TRACK_MAX_REQUESTS];
    double *out[MG_TRACK_MAX_REQUESTS];
    double h0, h1, px, py, pz, ref[3];
    int32_t align_m;
};

// MG_TRACK_CANDS candidates per workgroup: an eigenvector element read from L2 (the kernel's bound: NB x n_chan x L of them per
// workgroup) feeds that many fma chains.
#define MG_TRACK_BLOCK 128
#define MG_TRACK_CANDS 4
__device__ __forceinline__ void mg_joint_tracks_body(const mg_track_args &a, const int64_t block, unsigned char *smem) {
    const int ncp = a.NB * a.n_chan;
    double *cp_all = (double *)smem;                            // [CANDS][NB][n_chan] control points of the kept channels
    double *s_all = cp_all + (size_t)MG_TRACK_CANDS * ncp;      // [CANDS][L]
    double *al_all = s_all + (size_t)MG_TRACK_CANDS * a.L;      // [CANDS][8]: c, s, tx, tz, ty, aw, ay
    int *slot = (int *)(al_all + MG_TRACK_CANDS * 8);           // [D]
    const int64_t b0 = block * MG_TRACK_CANDS;
    const int tid = threadIdx.x, nc = a.n_chan, D = a.D, R = a.R, L = a.L;
    for (int e = tid; e < MG_TRACK_CANDS * L; e += MG_TRACK_BLOCK) {
        const int c = e / L, k = e - c * L;
        const int64_t bb = b0 + c < a.B ? b0 + c : a.B - 1;     // (a short last group repeats the last candidate; nothing of it is written)
        s_all[e] = a.lat_f64 ? ((const double *)a.lat)[bb * a.ld + k] : (double)((const float *)a.lat)[bb * a.ld + k];
    }
    for (int d = tid; d < D; d += MG_TRACK_BLOCK) slot[d] = a.slot[d];
    __syncthreads();
    for (int e = tid; e < ncp; e += MG_TRACK_BLOCK) {
        const int i = e / nc, q = e - i * nc;
        const int r = i * D + a.chan[q];
        double acc[MG_TRACK_CANDS];
#pragma unroll
        for (int c = 0; c < MG_TRACK_CANDS; c++) acc[c] = a.mean[r];
        for (int k = 0; k < L; k++) {
            const double ev = a.Et64[(size_t)k * R + r];
#pragma unroll
            for (int c = 0; c < MG_TRACK_CANDS; c++) acc[c] = fma(ev, s_all[c * L + k], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < MG_TRACK_CANDS; c++) cp_all[(size_t)c * ncp + e] = acc[c];
    }
    __syncthreads();
    if (tid < MG_TRACK_CANDS && a.align_mode != 0) {
        // the candidate's aligning transform from its FIRST control point (a clamped spline's value at t = 0)
        const double *cp = cp_all + (size_t)tid * ncp;
        double *al = al_all + tid * 8;
        auto ch0 = [&](int ch) { return cp[slot[ch]]; };
        const double p0x = ch0(0), p0z = ch0(2);
        double c, sn, ty;
        if (a.align_mode == 2) {
            c = a.h0; sn = a.h1; ty = a.py;
        } else {
            // MG_CONSTRAINT_VALUE_HEADING at t = 0 (mg_constraint_residual): the aligning node's global orientation applied to ref_dir, xz, unit
            double q[4], v[3];
            const double *rec = a.align_rec;
            double aw = 1.0, ax = 0.0, ay = 0.0, az = 0.0;
            for (int i = 0; i < a.align_m; i++) {
                const int qc = (int)rec[1 + 4 * i];
                double qw = ch0(qc), qx = ch0(qc + 1), qy = ch0(qc + 2), qz = ch0(qc + 3);
                const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
                qw *= inv; qx *= inv; qy *= inv; qz *= inv;
                const double nw = aw * qw - ax * qx - ay * qy - az * qz, nx = aw * qx + ax * qw + ay * qz - az * qy;
                const double ny = aw * qy - ax * qz + ay * qw + az * qx, nz = aw * qz + ax * qy - ay * qx + az * qw;
                aw = nw; ax = nx; ay = ny; az = nz;
            }
            q[0] = aw; q[1] = ax; q[2] = ay; q[3] = az;
            mg_rotate(q, a.ref[0], a.ref[1], a.ref[2], v);
            const double hx = v[0], hz = v[2];
            const double inv = 1.0 / sqrt(hx * hx + hz * hz);
            const double bx = 1.0 * hx * inv, bz = 1.0 * hz * inv;      // (weight 1 of the probing constraint)
            c = a.h0 * bx + a.h1 * bz;
            sn = a.h0 * bz - a.h1 * bx;
            ty = 0.0;
        }
        al[0] = c; al[1] = sn;
        al[2] = a.px - (c * p0x + sn * p0z);
        al[3] = a.pz - (c * p0z - sn * p0x);
        al[4] = ty;
        const double phi = atan2(sn, c);
        al[5] = cos(0.5 * phi); al[6] = sin(0.5 * phi);
    }
    __syncthreads();
    const bool aligned = a.align_mode != 0;
    const int n_here = a.B - b0 < MG_TRACK_CANDS ? (int)(a.B - b0) : MG_TRACK_CANDS;
    for (int rq = 0; rq < a.n_requests; rq++) {
        const int T = a.T[rq], j0 = a.req_joint0[rq], J = a.req_joint0[rq + 1] - j0;
        const int32_t *i0 = a.i0[rq];
        const double *w = a.w[rq];
        for (int ce = tid; ce < n_here * T * J; ce += MG_TRACK_BLOCK) {
            const int c = ce / (T * J), e = ce - c * (T * J);
            const double *cp = cp_all + (size_t)c * ncp, *al = al_all + c * 8;
            const double ac = al[0], as = al[1], tx = al[2], tz = al[3], ty = al[4], qaw = al[5], qay = al[6];
            double *out = a.out[rq] + (b0 + c) * (int64_t)T * J * 3;
            const int f = e / J, j = e - f * J;
            const double *wf = w + 4 * (size_t)f;
            const double *cf = cp + (size_t)i0[f] * nc;
            auto chan = [&](int ch) {       // the frame's channel: four taps of its control points (mg_back_project_frames_f64's statement)
                const double *cq = cf + slot[ch];
                double v = wf[0] * cq[0];
                v = fma(wf[1], cq[nc], v);
                v = fma(wf[2], cq[2 * nc], v);
                v = fma(wf[3], cq[3 * nc], v);
                return v;
            };
            double p0 = chan(0), p1 = chan(1), p2 = chan(2);
            if (aligned) {                  // mg_align_frames_kernel's statements
                const double x = p0, z = p2;
                p0 = ac * x + as * z + tx;
                p1 += ty;
                p2 = ac * z - as * x + tz;
            }
            const double *rec = a.records + (size_t)(j0 + j) * MG_TRACK_REC;
            const int m = (int)rec[0];
            double aw = 1.0, ax = 0.0, ay = 0.0, az = 0.0;
            for (int k = 0; k < m; k++) {   // mg_joint_positions_kernel's loop
                const int ch = (int)rec[1 + 4 * k];
                if (ch >= 0) {
                    double qw = chan(ch), qx = chan(ch + 1), qy = chan(ch + 2), qz = chan(ch + 3);
                    if (aligned && ch == 3 && D >= 7) {   // the root's quaternion turns with the candidate: (cos(phi / 2), 0, sin(phi / 2), 0) x q
                        const double w0 = qw, x0 = qx, y0 = qy, z0 = qz;
                        qw = qaw * w0 - qay * y0;
                        qx = qaw * x0 + qay * z0;
                        qy = qaw * y0 + qay * w0;
                        qz = qaw * z0 - qay * x0;
                    }
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
            out[(size_t)e * 3] = p0; out[(size_t)e * 3 + 1] = p1; out[(size_t)e * 3 + 2] = p2;
        }
    }
}

__global__ __launch_bounds__(MG_TRACK_BLOCK) void mg_joint_tracks_kernel(const mg_track_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mg_joint_tracks_body(a, blockIdx.x, smem);
}
// The tracks of SEVERAL options' candidates in one launch (a planner step: every option its own primitive, plan and candidates): the
// per-option arguments sit in a device table, workgroups [wg0[k], wg0[k + 1]) belong to option k.
__global__ __launch_bounds__(MG_TRACK_BLOCK) void mg_joint_tracks_multi_kernel(const mg_track_args *__restrict__ tab, const mg_opt_wgs w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int k = 0;
    while (k + 1 < w.n && (int)blockIdx.x >= w.wg0[k + 1]) k++;
    mg_joint_tracks_body(tab[k], (int64_t)blockIdx.x - w.wg0[k], smem);
}

extern "C" void mg_track_plan_destroy(mg_track_plan *pl) {
    if (!pl) return;
    if (pl->prim) { (void)hipSetDevice(pl->prim->ctx->device); (void)hipStreamSynchronize(pl->prim->ctx->stream); }
    (void)hipFree(pl->d_records); (void)hipFree(pl->d_align_rec); (void)hipFree(pl->d_chan); (void)hipFree(pl->d_slot);
    delete pl;
}

static int mg_track_chain_record(const char *who, const mg_skeleton_desc *sk, int joint, int n_dim, double *rec, std::vector<char> &need) {
    if (joint < 0 || joint >= sk->n_joints) { mg_set_error("%s: joint %d out of range", who, joint); return MG_ERR_INVALID_ARGUMENT; }
    std::vector<int> ch;
    for (int j = joint; j >= 0; j = sk->parents[j]) {
        if (sk->parents[j] >= j) { mg_set_error("%s: joint %d: parents must precede their children", who, j); return MG_ERR_INVALID_ARGUMENT; }
        ch.insert(ch.begin(), j);
    }
    const int m = (int)ch.size() - 1;
    if (m > MG_MAX_CHAIN) { mg_set_error("%s: chain of %d joints exceeds %d", who, m, MG_MAX_CHAIN); return MG_ERR_INVALID_ARGUMENT; }
    rec[0] = m;
    for (int k = 0; k < m; k++) {   // link k: rotation of chain joint k, offset of chain joint k + 1
        const int qc = sk->quat_channel[ch[(size_t)k]];
        if (qc >= 0 && qc + 4 > n_dim) { mg_set_error("%s: quaternion channel %d outside n_dim = %d", who, qc, n_dim); return MG_ERR_INVALID_ARGUMENT; }
        rec[1 + 4 * k] = qc;
        if (qc >= 0) for (int e = 0; e < 4; e++) need[(size_t)qc + e] = 1;
        for (int e = 0; e < 3; e++) rec[2 + 4 * k + e] = sk->offsets[(size_t)ch[(size_t)k + 1] * 3 + e];
    }
    return MG_OK;
}

// joints: the requests' joints back to back, n_joints[q] of them for request q (each 1 .. MG_FRAME_MAX_JOINTS).  align_joint: the
// node candidates are aligned through when an alignment is given at the call (0 = the root; -1: no alignment will ever be given).
extern "C" int mg_track_plan_create(mg_primitive *p, const mg_skeleton_desc *sk, int32_t n_requests, const int32_t *n_joints, const int32_t *joints,
                                    int32_t align_joint, mg_track_plan **out) {
    if (!p || !sk || !out || n_requests < 1 || n_requests > MG_TRACK_MAX_REQUESTS || !n_joints || !joints) {
        mg_set_error("mg_track_plan_create: bad arguments (1 .. %d requests)", MG_TRACK_MAX_REQUESTS);
        return MG_ERR_INVALID_ARGUMENT;
    }
    *out = nullptr;
    if (!(sk->n_joints > 0 && sk->parents && sk->offsets && sk->quat_channel && sk->parents[0] < 0) || p->D < 3) {
        mg_set_error("mg_track_plan_create: incomplete skeleton, or a primitive without root channels");
        return MG_ERR_INVALID_ARGUMENT;
    }
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    mg_track_plan *pl = new (std::nothrow) mg_track_plan();
    if (!pl) return MG_ERR_OUT_OF_MEMORY;
    pl->prim = p; pl->n_requests = n_requests;
    std::vector<char> need((size_t)p->D + 4, 0);
    need[0] = need[1] = need[2] = 1;
    int total = 0;
    for (int q = 0; q < n_requests; q++) {
        if (n_joints[q] < 1 || n_joints[q] > MG_FRAME_MAX_JOINTS) { mg_set_error("mg_track_plan_create: request %d has %d joints (1 .. %d)", q, n_joints[q], MG_FRAME_MAX_JOINTS); delete pl; return MG_ERR_INVALID_ARGUMENT; }
        pl->req_joint0[q] = total;
        total += n_joints[q];
    }
    pl->req_joint0[n_requests] = total;
    std::vector<double> records((size_t)total * MG_TRACK_REC, 0.0), arec(MG_TRACK_REC, 0.0);
    int rc = MG_OK;
    for (int o = 0; o < total && rc == MG_OK; o++) rc = mg_track_chain_record("mg_track_plan_create", sk, joints[o], p->D, &records[(size_t)o * MG_TRACK_REC], need);
    if (rc == MG_OK && align_joint >= 0) {
        // the aligning node's global orientation: the quaternions of the chain root .. node, the node's own included
        if (align_joint >= sk->n_joints) { mg_set_error("mg_track_plan_create: aligning joint %d out of range", align_joint); rc = MG_ERR_INVALID_ARGUMENT; }
        else {
            std::vector<int> ch;
            for (int j = align_joint; j >= 0; j = sk->parents[j]) ch.insert(ch.begin(), j);
            int m = 0;
            for (int j : ch) {
                const int qc = sk->quat_channel[j];
                if (qc < 0) continue;
                if (qc + 4 > p->D || m >= MG_MAX_CHAIN) { mg_set_error("mg_track_plan_create: aligning chain does not fit"); rc = MG_ERR_INVALID_ARGUMENT; break; }
                arec[1 + 4 * m] = qc;
                for (int e = 0; e < 4; e++) need[(size_t)qc + e] = 1;
                m++;
            }
            arec[0] = m;
            pl->align_m = m;
            pl->align_joint = align_joint;
        }
    }
    if (rc != MG_OK) { delete pl; return rc; }
    std::vector<int32_t> chan, slot((size_t)p->D, -1);
    for (int d = 0; d < p->D; d++)
        if (need[(size_t)d]) { slot[(size_t)d] = (int32_t)chan.size(); chan.push_back(d); }
    pl->n_chan = (int32_t)chan.size();
    auto up = [&](const void *h, size_t bytes, void **d) -> int {
        MG_HIP_CHECK(hipMalloc(d, std::max<size_t>(bytes, 16)));
        MG_HIP_CHECK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        return MG_OK;
    };
    rc = up(records.data(), records.size() * 8, (void **)&pl->d_records);
    if (rc == MG_OK) rc = up(arec.data(), arec.size() * 8, (void **)&pl->d_align_rec);
    if (rc == MG_OK) rc = up(chan.data(), chan.size() * 4, (void **)&pl->d_chan);
    if (rc == MG_OK) rc = up(slot.data(), slot.size() * 4, (void **)&pl->d_slot);
    if (rc != MG_OK) { mg_track_plan_destroy(pl); return rc; }
    *out = pl;
    return MG_OK;
}

// grids[q] (NULL = the canonical grid) and tracks_dev[q] (n, T_q, n_joints[q], 3) float64 per request; alignment NULL = local
// coordinates, else the record mg_constraint_set_create_aligned takes (joint must be the plan's align_joint, or MG_ALIGN_START_POSE).
static int mg_track_fill_args(const char *who, mg_track_plan *pl, const void *lat, int dt, int64_t B, int64_t ld, const mg_alignment_desc *al,
                              const mg_time_grid *const *grids, double *const *tracks_dev, mg_track_args *out, size_t *lds_out) {
    if (!pl || !pl->prim || !grids || !tracks_dev) { mg_set_error("%s: bad arguments", who); return MG_ERR_INVALID_ARGUMENT; }
    mg_primitive *p = pl->prim;
    if (B < 0 || B >= ((int64_t)1 << 31) || (dt != MG_F32 && dt != MG_F64) || ld < p->L) { mg_set_error("%s: bad batch (ld %lld < %d components?)", who, (long long)ld, p->L); return MG_ERR_INVALID_ARGUMENT; }
    if (B > 0 && !lat) { mg_set_error("%s: latents are NULL", who); return MG_ERR_INVALID_ARGUMENT; }
    mg_track_args a = {};
    a.Et64 = p->d_Et64; a.mean = p->d_mean; a.lat = lat; a.B = B; a.ld = ld; a.L = p->L; a.R = p->R; a.D = p->D; a.NB = p->NB; a.lat_f64 = dt == MG_F64 ? 1 : 0;
    a.n_requests = pl->n_requests; a.n_chan = pl->n_chan; a.records = pl->d_records; a.align_rec = pl->d_align_rec; a.chan = pl->d_chan; a.slot = pl->d_slot;
    a.align_m = pl->align_m;
    for (int q = 0; q <= pl->n_requests; q++) a.req_joint0[q] = pl->req_joint0[q];
    for (int q = 0; q < pl->n_requests; q++) {
        const mg_time_grid *g = grids[q] ? grids[q] : p->canonical;
        if (g->prim != p) { mg_set_error("%s: grid %d belongs to another primitive", who, q); return MG_ERR_INVALID_ARGUMENT; }
        if (!tracks_dev[q] || g->T < 1) { mg_set_error("%s: request %d has no output or an empty grid", who, q); return MG_ERR_INVALID_ARGUMENT; }
        a.T[q] = g->T; a.i0[q] = g->d_i0; a.w[q] = g->d_w; a.out[q] = tracks_dev[q];
    }
    a.align_mode = 0;
    if (al) {
        const double hn = std::sqrt(al->heading[0] * al->heading[0] + al->heading[1] * al->heading[1]);
        if (!(hn > 0.0) || !std::isfinite(hn)) { mg_set_error("%s: heading is zero or not finite", who); return MG_ERR_INVALID_ARGUMENT; }
        a.align_mode = al->joint == MG_ALIGN_START_POSE ? 2 : 1;
        if (a.align_mode == 1 && pl->align_m < 0) { mg_set_error("%s: the plan was made without an aligning joint", who); return MG_ERR_INVALID_ARGUMENT; }
        if (a.align_mode == 1 && al->joint != pl->align_joint) {
            mg_set_error("%s: the record aligns through joint %d, the plan was made for joint %d", who, al->joint, pl->align_joint);
            return MG_ERR_INVALID_ARGUMENT;
        }
        a.h0 = al->heading[0] / hn; a.h1 = al->heading[1] / hn; a.px = al->position[0]; a.py = al->position[1]; a.pz = al->position[2];
        for (int e = 0; e < 3; e++) a.ref[e] = al->ref_dir[e];
    }
    const size_t lds = (size_t)MG_TRACK_CANDS * ((size_t)p->NB * pl->n_chan + p->L + 8) * 8 + (size_t)p->D * 4 + 16;
    if (lds > 160 * 1024 - 64) { mg_set_error("%s: %d basis functions x %d channels do not fit LDS", who, p->NB, pl->n_chan); return MG_ERR_UNSUPPORTED; }
    *out = a;
    *lds_out = lds;
    return MG_OK;
}
static int mg_track_attributes(mg_context *ctx) {
    if (ctx->attr_traj & 4u) return MG_OK;
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_joint_tracks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_joint_tracks_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->attr_traj |= 4u;
    return MG_OK;
}

extern "C" int mg_joint_tracks(mg_track_plan *pl, const void *lat, int dt, int64_t B, int64_t ld, const mg_alignment_desc *al,
                               const mg_time_grid *const *grids, double *const *tracks_dev) {
    mg_track_args a;
    size_t lds = 0;
    { const int rc = mg_track_fill_args("mg_joint_tracks", pl, lat, dt, B, ld, al, grids, tracks_dev, &a, &lds); if (rc != MG_OK) return rc; }
    if (B == 0) return MG_OK;
    mg_primitive *p = pl->prim;
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    if (lds > 48 * 1024) { const int rc = mg_track_attributes(p->ctx); if (rc != MG_OK) return rc; }
    mg_prof_begin(p->ctx, 8);
    hipLaunchKernelGGL(mg_joint_tracks_kernel, dim3((unsigned)((B + MG_TRACK_CANDS - 1) / MG_TRACK_CANDS)), dim3(MG_TRACK_BLOCK), lds, p->ctx->stream, a);
    mg_prof_end(p->ctx, 8);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

// One planner step's per-frame constraint lists and first minima (mg_options_lists_kernel above), for options whose candidates and
// errors are on the device already (mg_options_step drew them and scored their keyframe constraints; mg_score_trajectories added the
// root trajectories): TWO launches whatever the number of options -- every option's joint tracks (mg_joint_tracks_multi_kernel), then
// every option's list + its first minimum -- and one read-back of the result records.
//   plans[k]: NULL for an option without a per-frame list (its first minimum is still taken); grids / tracks_dev: [n_options][MG_TRACK_MAX_REQUESTS]
//   (unused requests NULL); constraints: [n_options][MG_FRAME_LIST_MAX], n_constraints[k] of them used, request_of[k][i]: the plan's
//   request whose tracks constraint i reads; errors_dev[k] (n_samples) float64: read, added to, written back;
//   results_dev: record k at k * result_stride: {int64 index, float64 error, float64 latent[n_gmm_dims]}; results_host: NULL or where they are copied.
// The additions are mg_joint_tracks + mg_score_frame_constraints' per option, in the list's order: the same errors, the same winners.
extern "C" int mg_options_frame_lists(int32_t n_options, mg_primitive *const *prims, mg_track_plan *const *plans, const void *const *lat_dev, int dt, int64_t B,
                                      const int64_t *ld, const mg_alignment_desc *const *alignments, const mg_time_grid *const *grids, double *const *tracks_dev,
                                      const int32_t *n_constraints, const mg_frame_constraint_desc *const *constraints, const int32_t *request_of,
                                      double *const *errors_dev, void *results_dev, int64_t result_stride, void *results_host) {
    if (n_options < 1 || n_options > MG_OPT_LISTS_MAX || !prims || !lat_dev || !ld || !n_constraints || !errors_dev || !results_dev || B < 1 ||
        B >= ((int64_t)1 << 31) || (dt != MG_F32 && dt != MG_F64)) {
        mg_set_error("mg_options_frame_lists: bad arguments (1 .. %d options)", MG_OPT_LISTS_MAX);
        return MG_ERR_INVALID_ARGUMENT;
    }
    mg_context *ctx = prims[0] ? prims[0]->ctx : nullptr;
    for (int k = 0; k < n_options; k++) {
        if (!prims[k] || prims[k]->ctx != ctx || !lat_dev[k] || !errors_dev[k] || ld[k] < prims[k]->Lg || n_constraints[k] < 0 || n_constraints[k] > MG_FC_LIST_MAX ||
            result_stride < 16 + 8 * (int64_t)prims[k]->Lg || result_stride % 8 != 0) {
            mg_set_error("mg_options_frame_lists: option %d: NULL pointer, another context, ld / result_stride too small, or more than %d constraints", k, MG_FC_LIST_MAX);
            return MG_ERR_INVALID_ARGUMENT;
        }
        if (n_constraints[k] > 0 && (!plans || !plans[k] || plans[k]->prim != prims[k] || !grids || !tracks_dev || !constraints || !request_of)) {
            mg_set_error("mg_options_frame_lists: option %d has constraints but no plan of its primitive (or no grids / tracks / constraints)", k);
            return MG_ERR_INVALID_ARGUMENT;
        }
    }
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<mg_track_args> targs;
    std::vector<mg_opt_list_entry> lists((size_t)n_options);
    mg_opt_wgs tw = {}, lw = {};
    size_t t_lds = 0, l_lds = 0;
    bool tables_ok = true;
    const int wg_per = (int)((B + MG_FC_BLOCK - 1) / MG_FC_BLOCK), twg_per = (int)((B + MG_TRACK_CANDS - 1) / MG_TRACK_CANDS);
    for (int k = 0; k < n_options; k++) {
        mg_opt_list_entry &o = lists[(size_t)k];
        memset(&o, 0, sizeof(o));
        o.L.n = n_constraints[k]; o.L.accumulate = 1; o.L.B = B; o.L.out = errors_dev[k];
        o.x = lat_dev[k]; o.ld = ld[k]; o.x_f64 = dt == MG_F64 ? 1 : 0; o.Lw = prims[k]->Lg; o.result = (char *)results_dev + (size_t)k * result_stride;
        if (n_constraints[k] > 0) {
            mg_track_plan *pl = plans[k];
            mg_track_args a;
            size_t lds = 0;
            int rc = mg_track_fill_args("mg_options_frame_lists", pl, lat_dev[k], dt, B, ld[k], alignments ? alignments[k] : nullptr,
                                        grids + (size_t)k * MG_TRACK_MAX_REQUESTS, tracks_dev + (size_t)k * MG_TRACK_MAX_REQUESTS, &a, &lds);
            if (rc != MG_OK) return rc;
            targs.push_back(a);
            t_lds = std::max(t_lds, lds);
            tw.wg0[tw.n + 1] = tw.wg0[tw.n] + twg_per;
            tw.n++;
            for (int i = 0; i < n_constraints[k]; i++) {
                const int q = request_of[(size_t)k * MG_FC_LIST_MAX + i];
                if (q < 0 || q >= pl->n_requests) { mg_set_error("mg_options_frame_lists: option %d constraint %d reads request %d of %d", k, i, q, pl->n_requests); return MG_ERR_INVALID_ARGUMENT; }
                rc = mg_fc_args_from_desc("mg_options_frame_lists", prims[k], constraints[(size_t)k * MG_FC_LIST_MAX + i], a.out[q], B, a.T[q],
                                          pl->req_joint0[q + 1] - pl->req_joint0[q], nullptr, &o.L.c[i]);
                if (rc != MG_OK) return rc;
            }
            const size_t lds_l = mg_fc_place_tables(o.L.c, o.L.n);
            bool has_tables = false;
            for (int i = 0; i < o.L.n; i++)
                for (int j = 0; j < MG_FRAME_MAX_JOINTS; j++) has_tables = has_tables || o.L.c[i].traj[j].poly != nullptr;
            if (lds_l == 0 && has_tables) tables_ok = false;     // (one option's tables do not fit LDS: every option reads global memory)
            l_lds = std::max(l_lds, lds_l);
        }
        lw.wg0[k + 1] = lw.wg0[k] + wg_per;
    }
    lw.n = n_options;
    if (!tables_ok)
        for (auto &o : lists)
            for (int i = 0; i < o.L.n; i++)
                for (int j = 0; j < MG_FRAME_MAX_JOINTS; j++) o.L.c[i].traj[j].poly_off = o.L.c[i].traj[j].arc_off = -1;
    // the device tables, the partials and the counters: the context's, grown on demand
    const size_t t_bytes = targs.size() * sizeof(mg_track_args), l_bytes = lists.size() * sizeof(mg_opt_list_entry);
    const size_t p_bytes = (size_t)lw.wg0[n_options] * sizeof(mg_min_partial), c_off = (t_bytes + l_bytes + p_bytes + 255) / 256 * 256;
    const size_t need = c_off + MG_OPT_LISTS_MAX * sizeof(int32_t);
    if (ctx->lists_bytes < need) {
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->lists_dev) (void)hipFree(ctx->lists_dev);
        ctx->lists_dev = nullptr; ctx->lists_bytes = 0;
        const size_t want = std::max<size_t>(need, (size_t)256 << 10);
        MG_HIP_CHECK(hipMalloc(&ctx->lists_dev, want));
        MG_HIP_CHECK(hipMemset(ctx->lists_dev, 0, want));     // (the counters start at zero; their place moves with the sizes, so: all of it)
        ctx->lists_bytes = want; ctx->lists_counters_off = 0;
    }
    if (ctx->lists_counters_off != c_off) {                   // the counters' place moved: zero them there (stream ordered, before the launch)
        MG_HIP_CHECK(hipMemsetAsync((char *)ctx->lists_dev + c_off, 0, MG_OPT_LISTS_MAX * sizeof(int32_t), ctx->stream));
        ctx->lists_counters_off = c_off;
    }
    char *base = (char *)ctx->lists_dev;
    if (t_bytes) MG_HIP_CHECK(hipMemcpyAsync(base, targs.data(), t_bytes, hipMemcpyHostToDevice, ctx->stream));
    MG_HIP_CHECK(hipMemcpyAsync(base + t_bytes, lists.data(), l_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (tw.n > 0) {
        if (t_lds > 48 * 1024) { const int rc = mg_track_attributes(ctx); if (rc != MG_OK) return rc; }
        mg_prof_begin(ctx, 8);
        hipLaunchKernelGGL(mg_joint_tracks_multi_kernel, dim3((unsigned)tw.wg0[tw.n]), dim3(MG_TRACK_BLOCK), t_lds, ctx->stream, (const mg_track_args *)base, tw);
        mg_prof_end(ctx, 8);
        MG_HIP_CHECK(hipGetLastError());
    }
    if (l_lds && tables_ok) { const int rc = mg_fc_attributes(ctx); if (rc != MG_OK) return rc; }
    mg_prof_begin(ctx, 9);
    if (l_lds && tables_ok)
        hipLaunchKernelGGL(mg_options_lists_kernel<true>, dim3((unsigned)lw.wg0[n_options]), dim3(MG_FC_BLOCK), l_lds, ctx->stream,
                           (const mg_opt_list_entry *)(base + t_bytes), lw, (mg_min_partial *)(base + t_bytes + l_bytes), (int32_t *)(base + c_off));
    else
        hipLaunchKernelGGL(mg_options_lists_kernel<false>, dim3((unsigned)lw.wg0[n_options]), dim3(MG_FC_BLOCK), 0, ctx->stream,
                           (const mg_opt_list_entry *)(base + t_bytes), lw, (mg_min_partial *)(base + t_bytes + l_bytes), (int32_t *)(base + c_off));
    mg_prof_end(ctx, 9);
    MG_HIP_CHECK(hipGetLastError());
    if (results_host) {
        MG_HIP_CHECK(hipMemcpyAsync(results_host, results_dev, (size_t)n_options * result_stride, hipMemcpyDeviceToHost, ctx->stream));
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } else {
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));     // (the argument tables above were copied from this call's own host vectors)
    }
    return MG_OK;
}

"""Constraints that walk a joint's position through EVERY frame of a candidate's motion (reference
constraints/spatial_constraints/): trajectory constraints on joints other than the root, trajectory sets, discrete and local
trajectories, collision-avoidance position constraints -- and the local joint-rotation constraint, which reads one frame.

The fused keyframe scorer never materialises frames; these constraints need them.  The device does the heavy part for the
whole batch in three launches -- float64 back projection of every candidate, forward kinematics of the wanted joints in every
frame (`_capi.Primitive.joint_tracks`: mg_back_project_frames_f64 + mg_joint_positions), and, for trajectory constraints, the
monotone closest-point search over the tracks (mg_score_trajectory_points) -- and only (n, T, joints, 3) positions come back;
the constraints' own arithmetic (minimum over frames, arc-length look-ups, averages) is a few vectorised NumPy lines per
type, restated from the reference line by line.  The candidate's alignment to the previous motion (rotation about y and xz
translation per candidate, the same closed form the scorer applies) is applied to the tracks here.

Device forms (what candidate_scoring.constraints_to_device_form makes of the reference objects):
  {"type": "frame_joint_trajectory", "joint", "control_points", "min_u", "granularity", "weight"}
        TrajectoryConstraint on any joint -- trajectory_constraint.py:79-121
  {"type": "frame_ca_position", "joint", "target": [x | None, ...], "n_frames", "weight"}
        GlobalTransformCAConstraint -- keyframe_constraints/global_transform_ca_constraint.py:33-46
  {"type": "frame_discrete_trajectory", "joint", "points" (m, 3), "unconstrained": [axes], "weight"}
        DiscreteTrajectoryConstraint -- discrete_trajectory_constraint.py:66-90
  {"type": "frame_local_trajectory", "joint", "control_points", "granularity", "start_t", "n_frames", "weight"}
        LocalTrajectoryConstraint -- keyframe_constraints/local_trajectory_constraint.py:45-78
  {"type": "frame_trajectory_set", "joints", "trajectories": [{"control_points", "granularity", "range_start", "range_end"}],
   "arc_lengths", "n_frames", "weight"}
        TrajectorySetConstraint -- trajectory_set_constraint.py:82-104
  {"type": "frame_joint_rotation", "joint_index", "quaternion" (w, x, y, z), "frame_idx", "weight"}
        JointRotationConstraint -- keyframe_constraints/joint_rotation_constraint.py:55-72

PARITY UNPINNED where forward kinematics and alignment are (anim_utils); the target splines are pinned
(tests/golden/trajectory_spline.npz).  There is no CPU fallback: the tracks come from the GPU or the call raises.
"""
import numpy as np

from . import _capi
from .splines import CatmullRomPath

FRAME_TYPES = ("frame_joint_trajectory", "frame_ca_position", "frame_discrete_trajectory", "frame_local_trajectory",
               "frame_trajectory_set", "frame_joint_rotation")


def is_frame_constraint(c):
    return isinstance(c, dict) and c.get("type") in FRAME_TYPES


def split_frame_constraints(clist):
    """(the constraints the fused scorers take, the per-frame ones) of a device-form list"""
    return [c for c in clist if not is_frame_constraint(c)], [c for c in clist if is_frame_constraint(c)]


_ROOT_ONLY = None


def _skeleton_or_root(skeleton):
    global _ROOT_ONLY
    if skeleton is not None:
        return skeleton
    if _ROOT_ONLY is None:
        _ROOT_ONLY = _capi.Skeleton([("root", None, (0.0, 0.0, 0.0))], ["root"])
    return _ROOT_ONLY


def candidate_transforms(prim, S, skeleton, alignment):
    """The aligning transform of every candidate as the scorer derives it (mg_candidate_alignment): (cos, sin, tx, tz, ty),
    each (n,), or None in local coordinates.  The candidate's heading and root position in its FIRST control point come from
    the device (value constraints at t = 0: a clamped spline's first control point is its value there)."""
    if alignment is None:
        return None
    joint = alignment.get("joint", 0)
    ref_dir = tuple(alignment.get("ref_dir", (0.0, 0.0, 1.0)))
    start_pose = joint == _capi.MG_ALIGN_START_POSE
    probes = [{"type": "value_position", "t": 0.0, "weight": 1.0, "axis": 0}, {"type": "value_position", "t": 0.0, "weight": 1.0, "axis": 2}]
    if not start_pose:
        probes += [{"type": "value_heading", "t": 0.0, "weight": 1.0, "axis": 0, "joint": joint, "ref_dir": ref_dir},
                   {"type": "value_heading", "t": 0.0, "weight": 1.0, "axis": 2, "joint": joint, "ref_dir": ref_dir}]
    from .candidate_scoring import cached_constraint_set
    v = prim.score_constraint_residuals(cached_constraint_set(prim, probes, skeleton, None), S)
    p0x, p0z = v[:, 0], v[:, 1]
    h = np.asarray(alignment["heading"], dtype=np.float64)
    h = h / np.linalg.norm(h)
    if start_pose:
        c, s = np.full(len(v), h[0]), np.full(len(v), h[1])
        ty = np.full(len(v), float(alignment["position"][1]))
    else:
        bx, bz = v[:, 2], v[:, 3]
        c, s = h[0] * bx + h[1] * bz, h[0] * bz - h[1] * bx
        ty = np.zeros(len(v))
    tx = float(alignment["position"][0]) - (c * p0x + s * p0z)
    tz = float(alignment["position"][2]) - (c * p0z - s * p0x)
    return c, s, tx, tz, ty


def aligned_tracks(prim, S, skeleton, joints, alignment, times=None):
    """(n, T, len(joints), 3): the joints' global positions in every frame (times: canonical times, None = the canonical grid
    of get_motion_vector()), aligned like the candidate"""
    sk = _skeleton_or_root(skeleton)
    if skeleton is None:
        if any(j not in ("root", 0, None) for j in joints):
            raise NotImplementedError("per-frame constraints on joints %r need a skeleton (hip_skeleton); without one only the root's "
                                      "path (joint 0) exists" % (list(joints),))
        joints = [0] * len(joints)
    grid = None if times is None else prim.time_grid(np.asarray(times, dtype=np.float64))
    try:
        tr = prim.joint_tracks(sk, joints, S, grid)
    finally:
        if grid is not None:
            grid.close()
    tf = candidate_transforms(prim, S, skeleton, alignment)
    if tf is not None:
        c, s, tx, tz, ty = (a[:, None, None] for a in tf)
        x, z = tr[..., 0].copy(), tr[..., 2].copy()
        tr[..., 0] = c * x + s * z + tx
        tr[..., 2] = c * z - s * x + tz
        tr[..., 1] += ty
    return tr


def _masked_distance(target, p):
    """GlobalTransformConstraint._point_distance: axes whose target is None are ignored"""
    d2 = np.zeros(p.shape[:-1])
    for a in range(3):
        if target[a] is not None and not (isinstance(target[a], float) and np.isnan(target[a])):
            d2 = d2 + (float(target[a]) - p[..., a]) ** 2
    return np.sqrt(d2)


def frame_constraint_residuals(prim, S, c, skeleton=None, alignment=None):
    """The constraint's residual vector for every candidate, times its weight, and what MotionPrimitiveConstraints.evaluate adds
    for it: ((n, m) residuals, (n,) errors)."""
    S = np.asarray(S)
    n = len(S)
    w = float(c.get("weight", 1.0))
    kind = c["type"]
    F = prim.n_canonical_frames
    if kind == "frame_joint_trajectory":
        tr = aligned_tracks(prim, S, skeleton, [c["joint"]], alignment)[:, :, 0, :]
        from .candidate_scoring import cached_trajectory
        err, res = prim.score_trajectory_points(cached_trajectory(prim, {"type": "trajectory", "control_points": c["control_points"],
                                                                         "granularity": c.get("granularity", 1000)}),
                                                tr, c.get("min_u", 0.0), w, residuals=True)
        return res, err
    if kind == "frame_ca_position":
        nf = int(c.get("n_frames", F))
        tr = aligned_tracks(prim, S, skeleton, [c["joint"]], alignment, times=np.arange(nf, dtype=np.float64))[:, :, 0, :]
        err = w * _masked_distance(c["target"], tr).min(axis=1)            # the closest the joint ever comes (:37-38)
        return err[:, None], err
    if kind == "frame_discrete_trajectory":
        tr = aligned_tracks(prim, S, skeleton, [c["joint"]], alignment)[:, :, 0, :]
        pts = np.asarray(c["points"], dtype=np.float64)
        T, m = tr.shape[1], min(len(pts), tr.shape[1])
        keep = np.ones(3)
        for a in c.get("unconstrained", ()) or ():
            keep[int(a)] = 0.0
        res = np.zeros((n, T))
        res[:, :m] = np.linalg.norm((tr[:, :m] - pts[None, :m]) * keep, axis=2)
        return w * res, w * res.mean(axis=1)                               # np.average over all frames, zeros included (:66-90)
    if kind == "frame_local_trajectory":
        nf = int(c.get("n_frames", F))
        tr = aligned_tracks(prim, S, skeleton, [c["joint"]], alignment, times=np.arange(nf, dtype=np.float64))[:, :, 0, :]
        path = CatmullRomPath(c["control_points"], c.get("granularity", 1000))
        steps = np.linalg.norm(tr[:, 1:] - tr[:, :-1], axis=2)
        arc = float(c.get("start_t", 0.0)) + np.concatenate([np.zeros((n, 1)), np.cumsum(steps, axis=1)], axis=1)
        target = path.point_by_absolute_arc_length(arc.reshape(-1)).reshape(n, nf, -1)
        res = (target[..., 0] - tr[..., 0]) ** 2 + (target[..., 2] - tr[..., 2]) ** 2          # squared xz distance (:61-73)
        return w * res, w * res.sum(axis=1)
    if kind == "frame_trajectory_set":
        joints = list(c["joints"])
        nf = int(c.get("n_frames", F))
        tr = aligned_tracks(prim, S, skeleton, joints, alignment)[:, :nf]                          # frames of get_motion_vector()
        paths = [CatmullRomPath(t["control_points"], t.get("granularity", 1000)) for t in c["trajectories"]]
        steps = np.linalg.norm(tr[:, 1:] - tr[:, :-1], axis=3)                                      # (n, nf - 1, J)
        # the reference adds a frame's step AFTER it has looked the frame's targets up (:98-102): frames 0 and 1 use the initial
        # arc lengths, frame i >= 2 the path walked up to frame i - 1
        walked = np.concatenate([np.zeros((n, 2, len(joints))), np.cumsum(steps, axis=1)[:, :max(nf - 2, 0)]], axis=1)[:, :nf]
        arc = np.asarray(c.get("arc_lengths", np.zeros(len(joints))), dtype=np.float64)[None, None, :] + walked
        active = np.zeros((n, nf), dtype=bool)
        targets = np.empty_like(tr)
        for j, (path, t) in enumerate(zip(paths, c["trajectories"])):
            rs, re = t.get("range_start"), t.get("range_end")
            if rs is not None:
                active |= (arc[:, :, j] >= rs) & (arc[:, :, j] <= re)
            targets[:, :, j, :] = path.point_by_absolute_arc_length(arc[:, :, j].reshape(-1)).reshape(n, nf, -1)
        # np.average over the LIST of positions is the mean of all their components: a scalar centre (:93-96)
        res = np.abs(tr.reshape(n, nf, -1).mean(axis=2) - targets.reshape(n, nf, -1).mean(axis=2)) * active
        return w * res, w * res.mean(axis=1)                                                         # np.average of the n_canonical_frames entries
    if kind == "frame_joint_rotation":
        grid = prim.time_grid(np.array([float(c["frame_idx"])]))
        try:
            fr = prim.back_project_frames_f64(S, grid)[:, 0, :]
        finally:
            grid.close()
        ji = int(c["joint_index"])
        q = fr[:, 3 + 4 * ji:7 + 4 * ji].copy()
        if ji == 0:
            tf = candidate_transforms(prim, S, skeleton, alignment)
            if tf is not None:                                               # the root's quaternion turns with the candidate
                phi = np.arctan2(tf[1], tf[0])
                aw, ay = np.cos(0.5 * phi), np.sin(0.5 * phi)
                qw, qx, qy, qz = q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy(), q[:, 3].copy()
                q[:, 0], q[:, 1], q[:, 2], q[:, 3] = aw * qw - ay * qy, aw * qx + ay * qz, aw * qy + ay * qw, aw * qz - ay * qx
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        t = np.asarray(c["quaternion"], dtype=np.float64)
        t = t / np.linalg.norm(t)

        def rotmat(qq):
            ww, x, y, z = qq[..., 0], qq[..., 1], qq[..., 2], qq[..., 3]
            return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww),
                             2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww),
                             2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)], axis=-1)
        err = w * np.linalg.norm(rotmat(t)[None, :] - rotmat(q), axis=1)     # Frobenius norm of the matrix difference (:63-69)
        return err[:, None], err
    raise ValueError("unknown per-frame constraint %r" % (kind,))


def frame_constraints_errors(prim, S, frame_list, skeleton=None, alignment=None):
    """(n,) the sum of the per-frame constraints' errors, and their residual columns side by side"""
    total, blocks = np.zeros(len(S)), []
    for c in frame_list:
        res, err = frame_constraint_residuals(prim, S, c, skeleton, alignment)
        total = total + err
        blocks.append(res)
    return total, blocks

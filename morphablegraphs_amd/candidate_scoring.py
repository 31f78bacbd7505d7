"""Batched candidate scoring: the GPU replacement of the per-sample loop in
MotionPrimitiveGenerator.evaluate_samples_using_constraints
(reference morphablegraphs/motion_generator/motion_primitive_generator.py:230-261), shaped after the
MGRD precedent MGRDSampleFilter.score_samples (reference mgrd_sample_filter.py:63-75).

Global coordinates (`prev_frames` given, constraints not `is_local`): every candidate is first aligned to the last
previous frame as MotionPrimitiveConstraints.evaluate does (motion_primitive_constraints.py:110-114), on the device,
per candidate (mg_alignment_desc); the planner always scores this way (graph_walk_planner.py:179 never localises).

Constraints covered by the fused kernel: the FK-free ones path following uses
(locomotion_constraints_builder.py:82-117) -- root position at a canonical keyframe
(GlobalTransformConstraint, position only, root joint) and 2-D heading (Direction2DConstraint) -- and, given
a `_capi.Skeleton` (argument, or `hip_skeleton` attribute of the constraints object), the position of any
other joint by forward kinematics (hands, feet).  Trajectory constraints on the root joint
(trajectory_constraint.py:79-121: the root path against a spline over ALL frames) run in their own kernel
(mg_score_trajectory) and add to the same per-candidate error before the argmin.  Anything else raises -- there is no
silent CPU fallback.
"""
import collections

import numpy as np

from . import _capi

SAMPLING_MODE_GPU_BATCH = "gpu_batch"   # a 4th constrained_sampling_mode next to the reference's three


def constraints_to_device_form(constraints, root_joint=None):
    """Accepts either ready dicts {"type","t","weight","target"[,"ref_dir"]} or reference-shaped
    constraint objects: Direction2DConstraint (target_dir), GlobalTransformConstraint (position and / or
    orientation of a joint), RelativeTransformConstraint (position of a point given in a joint's frame),
    TwoHandConstraintSet (positions + joint_names), LookAtConstraint (target_position), FeetConstraint (left, right),
    PoseConstraint (pose_constraint + node_names + weights [+ velocity_constraint]).  One reference constraint may become
    several device constraints; "group" numbers the reference constraint's residual entry they add up to (a
    GlobalTransformConstraint is ONE residual = position error + orientation error, a TwoHandConstraint three,
    two_hand_constraint.py:66-74), see `group_residuals`."""
    out = []
    group = 0
    for c in constraints:
        if isinstance(c, dict):
            out.append(c)
            group += 1
            continue
        if hasattr(c, "joint_trajectories") and hasattr(c, "joint_names"):            # TrajectorySetConstraint
            out.append({"type": "frame_trajectory_set", "joints": list(c.joint_names), "weight": float(getattr(c, "weight_factor", 1.0)),
                        "trajectories": [dict(_spline_control_points(t), range_start=getattr(t, "range_start", None), range_end=getattr(t, "range_end", None))
                                         for t in c.joint_trajectories],
                        "arc_lengths": [float(v) for v in c.joint_arc_lengths], "n_frames": int(c.n_canonical_frames), "group": group})
            group += 1
            continue
        if hasattr(c, "point_list") and hasattr(c, "unconstrained_indices"):           # DiscreteTrajectoryConstraint
            out.append({"type": "frame_discrete_trajectory", "joint": c.joint_name, "points": [[float(v) for v in p] for p in c.point_list],
                        # target[None] = 0 zeroes the whole point (discrete_trajectory_constraint.py:81-82): no indices = no axis constrained
                        "unconstrained": [0, 1, 2] if c.unconstrained_indices is None else [int(i) for i in c.unconstrained_indices],
                        "weight": float(getattr(c, "weight_factor", 1.0)), "group": group})
            group += 1
            continue
        if getattr(c, "constraint_type", None) == "trajectory" or (hasattr(c, "min_arc_length") and hasattr(c, "full_arc_length")):
            out.append(trajectory_to_device_form(c, root_joint, group))
            group += 1
            continue
        if getattr(c, "constraint_type", None) == "local_trajectory":                  # LocalTrajectoryConstraint
            out.append(dict(_spline_control_points(c.trajectory), type="frame_local_trajectory", joint=c.joint_name, start_t=float(c.start_t),
                            n_frames=int(c.n_canonical_frames), weight=float(getattr(c, "weight_factor", 1.0)), group=group))
            group += 1
            continue
        if hasattr(c, "rotation_constraint") and hasattr(c, "frame_idx"):              # JointRotationConstraint
            names = list(c.skeleton.node_name_frame_map.keys())
            out.append({"type": "frame_joint_rotation", "joint_index": names.index(c.joint_name), "quaternion": _rotation_constraint_quaternion(c),
                        "frame_idx": float(c.frame_idx), "weight": float(getattr(c, "weight_factor", 1.0)), "group": group})
            group += 1
            continue
        if getattr(c, "constraint_type", None) == "ca_constraint":                      # GlobalTransformCAConstraint
            out.append({"type": "frame_ca_position", "joint": c.joint_name, "target": [None if v is None else float(v) for v in c.position],
                        "n_frames": int(c.n_canonical_frames), "weight": float(getattr(c, "weight_factor", 1.0)), "group": group})
            group += 1
            continue
        t = float(c.canonical_keyframe)
        w = float(getattr(c, "weight_factor", 1.0))
        if hasattr(c, "target_dir"):
            rd = getattr(getattr(c, "skeleton", None), "aligning_root_dir", (0.0, 0.0, 1.0))
            out.append({"type": "direction", "t": t, "weight": w, "target": [float(c.target_dir[0]), float(c.target_dir[1])],
                        "ref_dir": tuple(float(v) for v in rd), "group": group})
            group += 1
        elif hasattr(c, "positions") and hasattr(c, "joint_names"):          # TwoHandConstraint
            p0, p1 = np.asarray(c.positions[0], dtype=np.float64), np.asarray(c.positions[1], dtype=np.float64)
            center = p0 + 0.5 * (p1 - p0)
            out.append({"type": "joint_midpoint", "t": t, "weight": w, "target": [float(v) for v in center],
                        "joint": c.joint_names[0], "joint2": c.joint_names[1], "group": group})
            out.append({"type": "joint_position", "t": t, "weight": w, "target": [float(v) for v in p0], "joint": c.joint_names[0],
                        "group": group + 1})
            out.append({"type": "joint_position", "t": t, "weight": w, "target": [float(v) for v in p1], "joint": c.joint_names[1],
                        "group": group + 2})
            group += 3
        elif hasattr(c, "pose_constraint") and hasattr(c, "node_names"):          # PoseConstraint (pose_constraint.py:37-46)
            vel = getattr(c, "velocity_constraint", None)
            out.append({"type": "pose", "t": t, "weight": w, "joints": list(c.node_names),
                        "points": [[float(v) for v in p] for p in c.pose_constraint],
                        "weights": [float(v) for v in c.weights], "velocity": None if vel is None else [float(v) for v in vel],
                        "group": group})
            group += 1
        elif hasattr(c, "left") and hasattr(c, "right"):                          # FeetConstraint (feet_constraint.py:47-51)
            # its residuals carry weight_factor already and MotionPrimitiveConstraints.evaluate multiplies once more;
            # get_residual_vector_spline is the single entry [left + right]
            for joint, target in (("LeftFoot", c.left), ("RightFoot", c.right)):
                out.append({"type": "joint_position", "t": t, "weight": w * w, "target": [float(v) for v in target], "joint": joint,
                            "group": group})
            group += 1
        elif hasattr(c, "target_position") and not hasattr(c, "position"):      # LookAtConstraint (look_at_constraint.py:37-43)
            head = getattr(getattr(c, "skeleton", None), "head_joint", getattr(c, "joint_name", "Head"))
            out.append({"type": "look_at", "t": t, "weight": w, "target": [float(v) for v in c.target_position], "joint": head,
                        "group": group})
            group += 1
        elif getattr(c, "offset", None) is not None and getattr(c, "position", None) is not None:   # RelativeTransformConstraint
            out.append({"type": "joint_position", "t": t, "weight": w, "target": [float(v) for v in c.position],
                        "joint": c.joint_name, "offset": [float(v) for v in list(c.offset)[:3]], "group": group})
            group += 1
        elif getattr(c, "position", None) is not None or getattr(c, "orientation", None) is not None:
            joint = getattr(c, "joint_name", root_joint)
            root = root_joint if root_joint is not None else getattr(getattr(c, "skeleton", None), "root", None)
            n_before = len(out)
            if getattr(c, "position", None) is not None:
                if root is not None and joint is not None and joint != root:
                    # any other joint goes through the forward-kinematics constraint (needs a skeleton on the set)
                    out.append({"type": "joint_position", "t": t, "weight": w, "target": list(c.position), "joint": joint, "group": group})
                else:
                    out.append({"type": "position", "t": t, "weight": w, "target": list(c.position), "group": group})
            if getattr(c, "orientation", None) is not None:                  # (w, x, y, z), global_transform_constraint.py:53-58
                out.append({"type": "joint_orientation", "t": t, "weight": w, "orientation": [float(v) for v in c.orientation],
                            "joint": 0 if (joint is None or root is None or joint == root) else joint, "group": group})
            group += 1 if len(out) > n_before else 0
        else:
            raise NotImplementedError("constraint %r is not covered by the fused GPU scorer" % (type(c).__name__,))
    return out


def _spline_control_points(spline_owner):
    """{"control_points", "granularity"} of a reference ParameterizedSpline / TrajectoryConstraint (the padding of
    catmull_rom_spline.py:66-71 undone)"""
    spline = getattr(spline_owner, "spline", None)
    padded = getattr(spline, "control_points", None)
    if padded is None or not hasattr(spline, "_catmullrom_basematrix"):
        raise NotImplementedError("trajectory without a Catmull-Rom spline")
    return {"control_points": [list(map(float, p)) for p in padded[1:-2]], "granularity": int(getattr(spline_owner, "granularity", 1000))}


def _rotation_constraint_quaternion(c):
    """The wanted local rotation of a JointRotationConstraint as a unit quaternion (w, x, y, z): given as one, or as Euler angles in
    degrees, axes 'rxyz' (joint_rotation_constraint.py:41-52: rotations about the joint's own x, then y, then z)."""
    if c.rotation_type == "quaternion":
        q = np.asarray(c.rotation_constraint, dtype=np.float64)
        return (q / np.linalg.norm(q)).tolist()
    if c.rotation_type != "euler":
        raise ValueError("Unknown rotation type!")
    ax, ay, az = (np.deg2rad(float(v)) for v in c.rotation_constraint)

    def about(axis, angle):
        q = np.zeros(4)
        q[0], q[1 + axis] = np.cos(0.5 * angle), np.sin(0.5 * angle)
        return q

    def mul(a, b):
        return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])
    return mul(mul(about(0, ax), about(1, ay)), about(2, az)).tolist()       # rotating axes: R = Rx Ry Rz


def trajectory_to_device_form(c, root_joint=None, group=None):
    """A reference TrajectoryConstraint (trajectory_constraint.py:33-52: an AnnotatedSpline over control points with
    min_arc_length / full_arc_length, a joint and a weight) as the dict the device path takes.  Catmull-Rom splines only.  The root
    joint's path is scored inside the fused kernels ("trajectory"); any other joint's -- a collision-avoidance trajectory is one of
    those, the reference's optimiser evaluates it with the same method (graph_walk_optimizer.py:168-176) -- from the joint's
    forward-kinematics track ("frame_joint_trajectory")."""
    root = root_joint if root_joint is not None else getattr(getattr(c, "skeleton", None), "root", None)
    joint = getattr(c, "joint_name", root)
    if root is not None and joint is not None and joint != root:
        # another joint's path: forward kinematics over every frame (frame_constraints.py), the same search over its track
        full = float(c.full_arc_length)
        return dict(_spline_control_points(c), type="frame_joint_trajectory", joint=joint, min_u=float(c.min_arc_length) / full if full > 0.0 else 0.0,
                    weight=float(getattr(c, "weight_factor", 1.0)), group=group)
    spline = getattr(c, "spline", None)
    padded = getattr(spline, "control_points", None)
    if padded is None or not hasattr(spline, "_catmullrom_basematrix"):
        raise NotImplementedError("trajectory constraint without a Catmull-Rom spline")
    cps = [list(map(float, p)) for p in padded[1:-2]]          # the padding of catmull_rom_spline.py:66-71 undone
    full = float(c.full_arc_length)
    return {"type": "trajectory", "control_points": cps, "min_u": float(c.min_arc_length) / full if full > 0.0 else 0.0,
            "weight": float(getattr(c, "weight_factor", 1.0)), "granularity": int(getattr(c, "granularity", 1000)), "group": group}


def split_trajectories(clist):
    """(keyframe constraints, trajectory constraints) of a device-form list."""
    return [c for c in clist if c.get("type") != "trajectory"], [c for c in clist if c.get("type") == "trajectory"]


_TRAJ_CACHE = []   # [(key, _capi.Trajectory)]: a planner scores against the same trajectory for a whole action
_TRAJ_CACHE_SIZE = 64


def cached_trajectory(prim, c, pin=False):
    """The device trajectory for these control points, shared between calls.  A caller that keeps the object's HANDLE beyond the
    call at hand (a TrackScorer's constraint records, a planner step's side-by-side launch list) asks with pin=True and hands it
    back with release_trajectory(): a pinned trajectory is never destroyed under its holder -- when the cache lets go of it, the
    last release closes it (the records of a cached scorer pointed at trajectories the cache had closed)."""
    key = (prim.serial, prim.handle.value, _freeze(c["control_points"]), int(c.get("granularity", 1000)))
    t = None
    for i, (k, cand) in enumerate(_TRAJ_CACHE):
        if k == key and cand.handle:
            _TRAJ_CACHE.append(_TRAJ_CACHE.pop(i))
            t = cand
            break
    if t is None:
        t = _capi.Trajectory(prim, c["control_points"], c.get("granularity", 1000))
        t.pins, t.evicted = 0, False
        _TRAJ_CACHE.append((key, t))
    if pin:
        t.pins += 1
    while len(_TRAJ_CACHE) > _TRAJ_CACHE_SIZE:
        _drop_trajectory(_TRAJ_CACHE.pop(0)[1])
    return t


def _drop_trajectory(t):
    """the cache lets go of t: closed now, or by its last holder"""
    if getattr(t, "pins", 0) > 0:
        t.evicted = True
    else:
        t.close()


def release_trajectory(t):
    """the counterpart of cached_trajectory(..., pin=True)"""
    t.pins -= 1
    if t.pins <= 0 and t.evicted:
        t.close()


def _errors_with_trajectories_dev(prim, keyframe_list, trajectory_list, skeleton, alignment, d_S, dtype, n, ld, d_err):
    """sum of weighted constraint errors of n device-resident candidates into d_err (float64): the fused keyframe scorer,
    then one mg_score_trajectory launch per trajectory constraint adding to the same buffer."""
    if alignment is not None and trajectory_list and alignment.get("joint", 0) not in (0, _capi.MG_ALIGN_START_POSE):
        raise NotImplementedError("trajectory constraints in global coordinates need the root joint as aligning node")
    cset = cached_constraint_set(prim, keyframe_list, skeleton, alignment)
    prim.score_constraints_dev(cset, d_S, dtype, n, ld, d_err, np.float64)
    for c in trajectory_list:
        prim.score_trajectory_dev(cached_trajectory(prim, c), d_S, dtype, n, ld, d_err, c.get("min_u", 0.0), c.get("weight", 1.0),
                                  alignment, accumulate=True)


def group_residuals(clist, res):
    """(n, len(clist)) device residuals -> one column per reference residual entry: columns of the same "group"
    are summed (dicts without a group stand alone)."""
    groups, next_free = [], 0
    for c in clist:
        g = c.get("group") if isinstance(c, dict) else None
        groups.append(next_free if g is None else g)
        next_free = max(next_free, groups[-1]) + 1
    if len(set(groups)) == len(groups):
        return res
    order = sorted(set(groups))
    out = np.zeros((res.shape[0], len(order)), dtype=res.dtype)
    for col, g in enumerate(groups):
        out[:, order.index(g)] += res[:, col]
    return out


_CSET_CACHE = collections.OrderedDict()   # structure key -> ConstraintSet, most recent last: an optimizer calls the objective hundreds of times with
_CSET_CACHE_SIZE = 64   # the same constraints, and building a set uploads its fused keyframe matrices


_ROOT_ONLY = _capi.Skeleton([("root", None, (0.0, 0.0, 0.0))], ["root"])


def alignment_from_prev_frames(prev_frames, constraints=None, skeleton=None):
    """The alignment record for scoring in global coordinates, or None where the reference does not align:
    `is_local` constraints (motion_primitive_constraints.py:111) or no previous frames.  The aligning node and its
    reference direction come from the reference skeleton on the constraints (`skeleton.aligning_root_node`,
    `.aligning_root_dir`) when there is one, else the root joint and (0, 0, 1)."""
    if getattr(constraints, "is_local", False):
        return None
    if prev_frames is None:
        start_pose = getattr(constraints, "start_pose", None)
        if start_pose is not None and hasattr(constraints, "is_local"):
            return alignment_from_start_pose(start_pose)
        return None
    last = np.asarray(prev_frames, dtype=np.float64)
    last = last[-1] if last.ndim == 2 else last
    ref_sk = getattr(constraints, "skeleton", None)
    node = getattr(ref_sk, "aligning_root_node", None)
    ref_dir = tuple(float(v) for v in getattr(ref_sk, "aligning_root_dir", (0.0, 0.0, 1.0)))
    if skeleton is None:
        if node is not None and node != getattr(ref_sk, "root", node):
            raise NotImplementedError("aligning node %r is not the root joint: pass a _capi.Skeleton (hip_skeleton)" % (node,))
        return _ROOT_ONLY.alignment_to(last, 0, ref_dir)
    return skeleton.alignment_to(last, 0 if node is None else node, ref_dir)


def alignment_from_start_pose(start_pose):
    """The alignment record of the reference's start-pose branch (optimization/objective_functions.py:38-47): no previous
    frames, so every candidate is rotated by the start orientation and its first root position is moved in x and z.
    What the reference's arithmetic amounts to: `delta = start_pose["position"]` is the list inside the start pose, not a
    copy; m . first_root already contains that position, so delta[0] -= t_pos[0] leaves -(R first_root)[0] (and rewrites
    the start pose for the next call, where the same cancellation happens again): the first root position lands on
    x = z = 0 whatever the start position says, and heights are raised by its y.  That is what the device does; the start
    pose object is not touched.  get_transform_from_start_pose lives in anim_utils (absent, PARITY UNPINNED): Euler
    angles in degrees, of which only a rotation about y is covered."""
    orientation = [0.0 if v is None else float(v) for v in start_pose.get("orientation", (0.0, 0.0, 0.0))]
    if orientation[0] != 0.0 or orientation[2] != 0.0:
        raise NotImplementedError("start orientations with x or z angles are not aligned on the device")
    theta = np.radians(orientation[1])
    height = float(start_pose["position"][1])
    return {"joint": _capi.MG_ALIGN_START_POSE, "position": (0.0, height, 0.0), "heading": (float(np.cos(theta)), float(np.sin(theta)))}


def _freeze(v):
    """Nested tuples of plain Python values: hashable, and comparable with == / != whatever the caller handed over (arrays
    of any rank, NumPy scalars, nested lists)."""
    if isinstance(v, dict):
        return tuple(sorted((k, _freeze(x)) for k, x in v.items()))
    if isinstance(v, np.ndarray):
        v = v.tolist()
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(x) for x in v)
    if isinstance(v, np.generic):
        return v.item()
    return v


_STRUCT_FIELDS = ("joint", "joint2", "offset")
_VALUE_FIELDS = ("weight", "target", "ref_dir", "group")


def _flat(v):
    return _freeze(v) if isinstance(v, (list, tuple, np.ndarray, np.generic)) else v


def _structure_key(prim, clist, skeleton, alignment=None):
    """What a device set is built from and cannot change afterwards: per constraint its type, keyframe, joints and
    relative point, plus the aligning joint.  Targets, weights, reference vectors and the previous frame are values
    (ConstraintSet.update)."""
    # (a pose constraint's cloud is part of the set's tables: all of it is structure)
    items = tuple((c["type"], float(c["t"]), _flat(c.get("joint")), _flat(c.get("joint2")), _flat(c.get("offset")),
                   _freeze(c) if c["type"] == "pose" else None) for c in clist)
    # (a Skeleton carries a serial number: id() of a collected one can be handed to a new object)
    sk = None if skeleton is None else getattr(skeleton, "serial", id(skeleton))
    return (prim.serial, prim.handle.value, sk, items, None if alignment is None else _flat(alignment.get("joint", 0)))


def _values_key(clist, alignment):
    """The values of a set with a given structure (what ConstraintSet.update rewrites)."""
    vals = tuple((c.get("weight"), _flat(c.get("target")), _flat(c.get("ref_dir"))) for c in clist)
    if alignment is None:
        return (vals, None)
    return (vals, tuple((k, _flat(alignment[k])) for k in sorted(alignment)))


# bumped whenever a shared set is created, rewritten or closed: "nothing happened to any set since" is one integer comparison
# (the planner step's whole-step shortcut, motion_state_graph.HipPrimitiveSet.evaluate_options_on_device)
CSET_GENERATION = [0]


def cached_constraint_set(prim, clist, skeleton=None, alignment=None):
    """A device constraint set for these (device-form) constraints, reused across calls.  An optimizer evaluates the
    same constraints again and again (same values: nothing to do); a planner scores the same KIND of constraints with
    new goals and a new previous frame every step (same structure: the values are rewritten by one small launch,
    ConstraintSet.update, instead of building a new set, which costs about 200 us).  The returned set is SHARED: use
    it for the call at hand and ask again next time -- a later request with the same structure and other values
    rewrites it in place (stream ordered, so launches already enqueued keep the values they were enqueued with)."""
    key = _structure_key(prim, clist, skeleton, alignment)
    values = _values_key(clist, alignment)
    cs = _CSET_CACHE.get(key)
    if cs is not None and not (cs.handle and cs.prim.handle and cs.prim.ctx.handle):   # its primitive has been closed meanwhile
        del _CSET_CACHE[key]
        CSET_GENERATION[0] += 1
        cs = None
    if cs is not None:
        _CSET_CACHE.move_to_end(key)
        if cs.cached_values != values:
            cs.update(clist, alignment)
            cs.cached_values = values
            CSET_GENERATION[0] += 1
        return cs
    cs = _capi.ConstraintSet(prim, clist, skeleton, alignment)
    cs.cached_values = values
    _CSET_CACHE[key] = cs
    CSET_GENERATION[0] += 1
    while len(_CSET_CACHE) > _CSET_CACHE_SIZE:
        _CSET_CACHE.popitem(last=False)[1].close()
    return cs


def clear_constraint_cache():
    """Drop the cached device constraint sets (call before closing a primitive they belong to)."""
    CSET_GENERATION[0] += 1
    while _CSET_CACHE:
        _CSET_CACHE.popitem()[1].close()
    while _TRAJ_CACHE:
        _drop_trajectory(_TRAJ_CACHE.pop()[1])



class HipSampleFilter(object):
    """score_samples(primitive, samples, constraints) -> errors[n], like MGRDSampleFilter."""

    @staticmethod
    def score_samples(motion_primitive, samples, constraints, dtype=np.float64, skeleton=None, prev_frames=None):
        prim = motion_primitive._prim if hasattr(motion_primitive, "_prim") else motion_primitive
        clist = constraints.constraints if hasattr(constraints, "constraints") else constraints
        device_form = constraints_to_device_form(clist)
        alignment = alignment_from_prev_frames(prev_frames, constraints, skeleton)
        from .frame_constraints import is_frame_constraint
        if any(is_frame_constraint(c) for c in device_form):
            return errors_of_samples(prim, device_form, skeleton, alignment, samples).astype(dtype)
        keyframes, trajectories = split_trajectories(device_form)
        if trajectories:
            S = _capi._latents(samples)
            d_S, d_e = prim.ctx.upload(S), prim.ctx.malloc(max(len(S), 1) * 8)
            try:
                _errors_with_trajectories_dev(prim, keyframes, trajectories, skeleton, alignment, d_S, S.dtype, len(S), S.shape[1], d_e)
                return prim.ctx.download(d_e, (len(S),), np.float64).astype(dtype)
            finally:
                d_S.free()
                d_e.free()
        cset = _capi.ConstraintSet(prim, keyframes, skeleton, alignment)
        try:
            return prim.score_constraints(cset, np.asarray(samples), dtype=dtype)
        finally:
            cset.close()


def _prim_of(mp_node):
    prim_obj = mp_node.motion_primitive if hasattr(mp_node, "motion_primitive") else mp_node
    return prim_obj, prim_obj._prim


def all_errors_dev(prim, device_form, skeleton, alignment, S, d_S, d_err):
    """The sum of ALL constraints' weighted errors of candidates S (host latents, d_S their copy on the device) into d_err (n,)
    float64 on the device: keyframe and root-trajectory constraints by the fused scorers, then the per-frame constraints
    (frame_constraints.py) adding to the same buffer."""
    from .frame_constraints import split_frame_constraints, add_frame_constraints_dev
    fused, frames = split_frame_constraints(device_form)
    keyframes, trajectories = split_trajectories(fused)
    _errors_with_trajectories_dev(prim, keyframes, trajectories, skeleton, alignment, d_S, S.dtype, len(S), S.shape[1], d_err)
    if frames:
        add_frame_constraints_dev(prim, S, frames, skeleton, alignment, d_err, accumulate=True)


def errors_of_samples(prim, device_form, skeleton, alignment, samples):
    """(n,) float64: the sum of ALL constraints' weighted errors for host-resident candidates."""
    S = _capi._latents(samples)
    if len(S) == 0:
        return np.zeros(0)
    d_S, d_e = prim.ctx.upload(S), prim.ctx.malloc(len(S) * 8)
    try:
        all_errors_dev(prim, device_form, skeleton, alignment, S, d_S, d_e)
        return prim.ctx.download(d_e, (len(S),), np.float64)
    finally:
        d_S.free()
        d_e.free()


def first_minimum_of_block(mp_node, device_form, alignment, samples, skeleton=None):
    """(index, error) of the first minimum among `samples` (n, L) for constraints already in device form: what one rank does
    with its block of a sharded evaluate_samples_using_constraints, and the whole of the single-GPU call."""
    prim_obj, prim = _prim_of(mp_node)
    from .frame_constraints import is_frame_constraint
    if len(samples) == 0:
        return 0, float("inf")
    if any(is_frame_constraint(c) for c in device_form):       # per-frame constraints add to the fused scorers' sum on the device
        S = _capi._latents(samples)
        d_S, d_e = prim.ctx.upload(S), prim.ctx.malloc(len(S) * 8)
        try:
            all_errors_dev(prim, device_form, skeleton, alignment, S, d_S, d_e)
            return prim.ctx.argmin_first(d_e, len(S), np.float64)
        finally:
            d_S.free()
            d_e.free()
    keyframes, trajectories = split_trajectories(device_form)
    if trajectories:
        S = _capi._latents(samples)
        d_S, d_e = prim.ctx.upload(S), prim.ctx.malloc(max(len(S), 1) * 8)
        try:
            _errors_with_trajectories_dev(prim, keyframes, trajectories, skeleton, alignment, d_S, S.dtype, len(S), S.shape[1], d_e)
            return prim.ctx.argmin_first(d_e, len(S), np.float64)
        finally:
            d_S.free()
            d_e.free()
    cset = cached_constraint_set(prim, keyframes, skeleton, alignment)
    return prim.best_candidate(cset, samples)   # one upload, two launches, 16 bytes back


def sample_rows_and_first_minimum(mp_node, device_form, alignment, counts, seed, row_begin, row_end, skeleton=None, dtype=np.float32):
    """Rows [row_begin, row_end) of the device sampler's draw for (counts, seed), scored, first minimum: returns
    (index relative to row_begin, error, winning latent at full width).  One rank's block of a sharded gpu_batch step; with
    (0, n) the whole single-GPU step.  Only the winner leaves the GPU."""
    prim_obj, prim = _prim_of(mp_node)
    ctx = prim.ctx
    from .frame_constraints import is_frame_constraint
    keyframes, trajectories = split_trajectories(device_form)
    L = prim.n_gmm_dims          # the full sample (spatial + time latents); scoring reads its first n_components columns
    m = int(row_end) - int(row_begin)
    item = np.dtype(dtype).itemsize
    d_x = ctx.malloc(max(m, 1) * L * item)
    try:
        prim.gmm_sample_dev(np.asarray(counts, dtype=np.int64), seed, d_x, dtype, L, rows=(int(row_begin), m))
        if any(is_frame_constraint(c) for c in device_form):   # per-frame constraints: frames and joint tracks of the drawn rows
            X = ctx.download(d_x, (m, L), dtype)                 # (the batch object keeps its own copy of the latents)
            d_e = ctx.malloc(max(m, 1) * 8)
            try:
                all_errors_dev(prim, device_form, skeleton, alignment, X, d_x, d_e)
                best_idx, min_error = ctx.argmin_first(d_e, m, np.float64)
            finally:
                d_e.free()
            return best_idx, min_error, X[best_idx].astype(np.float64)
        if trajectories:
            d_e = ctx.malloc(max(m, 1) * 8)
            try:
                _errors_with_trajectories_dev(prim, keyframes, trajectories, skeleton, alignment, d_x, dtype, m, L, d_e)
                best_idx, min_error = ctx.argmin_first(d_e, m, np.float64)
            finally:
                d_e.free()
        else:
            cset = cached_constraint_set(prim, keyframes, skeleton, alignment)
            best_idx, min_error = prim.best_candidate_dev(cset, d_x, dtype, m, L)
        best = ctx.download(d_x.ptr.value + best_idx * L * item, (L,), dtype)
    finally:
        d_x.free()
    return best_idx, min_error, best.astype(np.float64)


def _node_key(mp_node, communicator):
    """how the ranks of a sharded call name the primitive: the key under which every rank holds it"""
    key = getattr(mp_node, "node_key", None)
    if key is None:
        prim_obj, _ = _prim_of(mp_node)
        key = getattr(prim_obj, "name", None)
    return key


def evaluate_samples_using_constraints(samples, mp_node, constraints, prev_frames=None, skeleton=None, communicator=None):
    """Drop-in for MotionPrimitiveGenerator.evaluate_samples_using_constraints: returns (samples[best_idx],
    min_error) with the reference's first-minimum rule, and updates constraints.min_error / constraints.evaluations
    when those attributes exist.  With `prev_frames` (and constraints that are not `is_local`) every candidate is
    aligned to the previous motion first, like MotionPrimitiveConstraints.evaluate.
    communicator (distributed.MgCommunicator / FileCommunicator, rank 0 calling, the other ranks in distributed.worker_loop):
    the candidates travel to the ranks with the command, every rank scores its contiguous block, one all-gather of the
    ranks' first minima -- the same winner as the single-GPU call."""
    samples = np.asarray(samples)
    clist = constraints.constraints if hasattr(constraints, "constraints") else constraints
    skeleton = skeleton if skeleton is not None else getattr(constraints, "hip_skeleton", None)
    device_form = constraints_to_device_form(clist)
    alignment = alignment_from_prev_frames(prev_frames, constraints, skeleton)
    if communicator is not None and communicator.world > 1:
        from . import distributed
        best_idx, min_error, _ = distributed.run_command(communicator, {_node_key(mp_node, communicator): mp_node, "__skeleton__": skeleton},
                                                         {"op": "evaluate_samples", "node": _node_key(mp_node, communicator), "samples": samples,
                                                          "constraints": device_form, "alignment": alignment, "skeleton": skeleton is not None})
    else:
        best_idx, min_error = first_minimum_of_block(mp_node, device_form, alignment, samples, skeleton)
    if hasattr(constraints, "min_error"):
        constraints.min_error = min_error
    if hasattr(constraints, "evaluations"):
        constraints.evaluations += len(samples)
    return samples[best_idx], min_error


def sample_and_evaluate_on_device(mp_node, constraints, n_samples, seed, skeleton=None, dtype=np.float32, prev_frames=None, communicator=None):
    """The gpu_batch step without the host round trip: component counts from NumPy's global stream (the first
    draw sklearn's GaussianMixture.sample makes), latents from the device Philox sampler (NOT sklearn's Mersenne
    stream: distributional parity only), scoring and first-minimum argmin on the device; only the winning latent
    vector comes back.  Returns (best_sample, min_error).
    communicator: rank 0 broadcasts (constraint values, counts, seed); every rank draws ITS rows of the one draw (the
    generator is counter based: the union of the ranks' blocks is the single-GPU draw, bit for bit), scores them and offers
    its first minimum; the same (best_sample, min_error) as without a communicator."""
    prim_obj, prim = _prim_of(mp_node)
    clist = constraints.constraints if hasattr(constraints, "constraints") else constraints
    skeleton = skeleton if skeleton is not None else getattr(constraints, "hip_skeleton", None)
    device_form = constraints_to_device_form(clist)
    alignment = alignment_from_prev_frames(prev_frames, constraints, skeleton)
    weights = np.asarray(prim_obj.gaussian_mixture_model.weights_, dtype=np.float64)
    counts = np.random.multinomial(int(n_samples), weights / weights.sum()).astype(np.int64)
    if communicator is not None and communicator.world > 1:
        from . import distributed
        _, min_error, best = distributed.run_command(communicator, {_node_key(mp_node, communicator): mp_node, "__skeleton__": skeleton},
                                                     {"op": "sample_and_evaluate", "node": _node_key(mp_node, communicator), "constraints": device_form,
                                                      "alignment": alignment, "skeleton": skeleton is not None, "counts": counts, "seed": int(seed),
                                                      "dtype": np.dtype(dtype).name})
        best = best.astype(dtype).astype(np.float64)
    else:
        _, min_error, best = sample_rows_and_first_minimum(mp_node, device_form, alignment, counts, seed, 0, int(n_samples), skeleton, dtype)
    if hasattr(constraints, "min_error"):
        constraints.min_error = min_error
    if hasattr(constraints, "evaluations"):
        constraints.evaluations += int(n_samples)
    return best, min_error
